// dense_aggregate_c16.hip -- the dense-parcellation aggregate (dense_aggregate_h16.hip) with the
// operator stored per MFMA fragment in the form that fits it (BASELINE config 5: 1000-ROI graphs,
// 10 % density, fp16 storage).
//
//   Y_g = M_g X_g on the fp16 matrix cores, as in dense_aggregate_h16.hip.  The dense operator is
//   2 bytes x P^2 per graph -- 134 MB per 64 x 1000-ROI batch -- and the kernel is HBM-bound on
//   exactly those bytes (84 us from cold caches).  Small-world connectomes are bimodal: the
//   fragments (32 rows x 16 sources) on the lattice band are nearly full, the rest hold a handful
//   of rewired edges.  So every fragment of a (graph, row block) goes to one of two lists:
//     dense list  : fragments with more than 64 non-zeros, 1 KB each, MFMA-operand-major as in M;
//     sparse list : fragments with 1..64 non-zeros as one chunk of 64 entries
//                   (slot in the fragment | half value << 16, padding slot 0xFFFF; four chunks
//                   interleaved per lane so that one 16-byte load brings four);
//     empty fragments are in neither.  Each list carries the k-step of its items.
//   A wave streams its row block's dense list straight into MFMA operands, then rebuilds the
//   sparse fragments in a private 1 KB LDS slab: one 2-byte store per entry (one per lane), one
//   16-byte read in operand layout, one 2-byte store per entry to clear the slab again.  Same
//   MFMAs as the dense kernel, accumulated dense-list-first: equal to it up to the order of the
//   fp32 accumulation (bit-identical when a row block's fragments are all of one kind).
//
//   cgnn_dense_pack_count / cgnn_dense_pack_fill build the lists from a CSR ordering; duplicate
//   edges add up in fp32 before the one rounding to half, exactly as cgnn_dense_adj_f16 builds M.
#include <hip/hip_fp16.h>
#include "common.h"
#include "drop_ew.h"

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int C_NW = 12;
constexpr int C_THR = C_NW * 64;
constexpr int C_MAXP = 1024;
constexpr int C_KPAD = 24;                     // halves of padding per transposed row
constexpr int C_DAHEAD = 8;                    // dense fragments in flight per wave
constexpr int C_SAHEAD = 4;                    // 16-byte loads (4 sparse chunks each) in flight per wave
constexpr uint32_t C_PAD = 0xFFFFu;            // slot of a padding entry
constexpr uint32_t C_DENSE = 0x80000000u;      // fpos flag: the fragment goes to the dense list
constexpr uint32_t C_EMPTY = 0xFFFFFFFFu;      // fpos of an empty fragment

__host__ __device__ inline int c_kp(int P) { return (P + 15) / 16 * 16 + C_KPAD; }

// ------------------------------------------------------------------ builder (count / fill)
// one block per (row block, graph): the 32 rows accumulate in LDS (serial per row, COO order:
// reproducible), then every k-step's 512 slots are counted / written in slot order.
template <bool FILL>
__global__ void __launch_bounds__(256) k_dense_pack(
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
    const float* __restrict__ coef, const float* __restrict__ selfc,
    const int32_t* __restrict__ gptr, int P, uint32_t* __restrict__ counts,
    const uint32_t* __restrict__ fpos, __half* __restrict__ dfrag, int32_t* __restrict__ dstep,
    uint32_t* __restrict__ sent, int32_t* __restrict__ sstep) {
  extern __shared__ float rows[];                 // [32][P]
  __shared__ int wc[2][4];
  const int t = threadIdx.x, rb = blockIdx.x, g = blockIdx.y;
  const int base = gptr[g], n = gptr[g + 1] - base;
  const int S = P >> 4, NRB = P >> 5;
  for (int i = t; i < 32 * P; i += 256) rows[i] = 0.f;
  __syncthreads();
  if (t < 32) {
    const int d = 32 * rb + t;
    if (d < n) {
      const int r = base + d;
      for (int e = rowptr[r]; e < rowptr[r + 1]; ++e) {
        const int s = col[e] - base;
        if (s >= 0 && s < n) rows[t * P + s] += coef[e];
      }
      if (selfc) rows[t * P + d] += selfc[r];
    }
  }
  __syncthreads();
  const int lane = t & 63, w = t >> 6;
  const int64_t frag0 = ((int64_t)g * NRB + rb) * S;
  for (int s = 0; s < S; ++s) {
    uint32_t bits[2];
    bool nz[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int slot = t + 256 * u, ln = slot >> 3, e = slot & 7;
      const int i = ln & 31, k = 16 * s + 8 * (ln >> 5) + e;
      bits[u] = __half_as_ushort(__float2half_rn(rows[i * P + k]));
      nz[u] = (bits[u] & 0x7FFFu) != 0u;
    }
    if (!FILL) {
      const unsigned long long b0 = __ballot(nz[0]), b1 = __ballot(nz[1]);
      if (lane == 0) wc[0][w] = __popcll(b0) + __popcll(b1);
      __syncthreads();
      if (t == 0) counts[frag0 + s] = (uint32_t)(wc[0][0] + wc[0][1] + wc[0][2] + wc[0][3]);
      __syncthreads();
      continue;
    }
    const uint32_t fp = fpos[frag0 + s];            // block-uniform
    if (fp == C_EMPTY) continue;
    if (fp & C_DENSE) {
      const int64_t pos = fp & ~C_DENSE;
      // slots t and t + 256 of the operand-major fragment: [lane][8 halves]
      reinterpret_cast<unsigned short*>(dfrag)[pos * 512 + t] = (unsigned short)bits[0];
      reinterpret_cast<unsigned short*>(dfrag)[pos * 512 + 256 + t] = (unsigned short)bits[1];
      if (t == 0) dstep[pos] = s;
      continue;
    }
    const unsigned long long b0 = __ballot(nz[0]), b1 = __ballot(nz[1]);
    if (lane == 0) { wc[0][w] = __popcll(b0); wc[1][w] = __popcll(b1); }
    __syncthreads();
    int total0 = 0, pre0 = 0, pre1 = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (k < w) { pre0 += wc[0][k]; pre1 += wc[1][k]; }
      total0 += wc[0][k];
    }
    // entry i (< 64) of chunk fp: [chunk/4][lane = i][chunk%4]; the tail keeps its padding
    uint32_t* out = sent + ((((int64_t)(fp >> 2)) * 64) << 2) + (fp & 3u);
    const unsigned long long below = (1ull << lane) - 1ull;
    if (nz[0]) out[(pre0 + __popcll(b0 & below)) << 2] = (uint32_t)t | (bits[0] << 16);
    if (nz[1]) out[(total0 + pre1 + __popcll(b1 & below)) << 2] = (uint32_t)(t + 256) | (bits[1] << 16);
    if (t == 0) sstep[fp] = s;
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------ Y_g = M_g X_g
// BNB (backward of a GCN layer: dT = A_hat^T dY): X is not read but FORMED while the slice is staged,
//   dY = a * (dX' * f - c1 - xhat * c2),  f = act' * keep / (1-p),  xhat = (Yl - mean) * invstd
// from the gradient of the layer's activation (dX', or the readout's dP[graph] / n_g for the last
// layer), the layer's pre-BatchNorm output Yl, its keep bytes, coefficient block and the backward
// coefficients c1|c2 -- the arithmetic of k_bn_act_apply<true> (elementwise.hip), rounded to half as the
// stored dY would be.  dY feeds nothing but this product (and db = column sums of dY, left per graph in
// cs_slab [B][F] fp64), so the apply pass, its write of dY and this kernel's read of it disappear.
struct DenseBnBwd {
  const __half* dX;          // [M][ldx] or NULL (then dP)
  const float* dP;           // [B][F] readout gradient (last layer) or NULL
  const __half* Yl;          // [M][ldy]
  int64_t ldyl;
  const uint8_t* mask;       // [M][F/4] or NULL
  const float* coef;         // [a | b | mean | invstd] x F
  const float* bwc;          // [c1 | c2] x F
  int relu;
  float scale;
  double* cs_slab;           // [B][F]
};

template <bool BNB>
__global__ void __launch_bounds__(C_THR) k_dense_agg_c(
    const __half* __restrict__ dfrag, const int32_t* __restrict__ dstep, const uint32_t* __restrict__ doff,
    const uint32_t* __restrict__ sent, const int32_t* __restrict__ sstep, const uint32_t* __restrict__ soff,
    int P, const int32_t* __restrict__ gptr, int B, const __half* __restrict__ X, int64_t ldx,
    int nslices, const float* __restrict__ bias, __half* __restrict__ Y, int64_t ldy,
    double* __restrict__ stat_slab, int rparts, DenseBnBwd bb) {
  extern __shared__ __attribute__((aligned(16))) __half Xt[];       // [64][KP], the fragment slabs, wacc
  const int KP = c_kp(P);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  __half* frag2 = Xt + 64 * KP + wave * 1024;                        // this wave's two [64 lanes][8] slabs
  const int NRB = P >> 5;
  // a graph's work = nslices column slices x rparts runs of row blocks (rparts > 1 when the batch has
  // fewer (graph, slice) pairs than the chip has CUs -- layer 0's single 64-column panel of a 64-graph
  // batch would otherwise occupy a quarter of them; the parts re-stage the same slice from L2)
  const int sub = nslices * rparts;
  const int units = B * sub;
  // the slabs start (and are always left) all zero
  *reinterpret_cast<uint4*>(frag2 + 8 * lane) = make_uint4(0u, 0u, 0u, 0u);
  *reinterpret_cast<uint4*>(frag2 + 512 + 8 * lane) = make_uint4(0u, 0u, 0u, 0u);
  // BatchNorm statistics of the (half-rounded) output, optional: per-wave partials go through the
  // slab area (zero again afterwards), the workgroup's running column sums sit behind it
  double* red = reinterpret_cast<double*>(Xt + 64 * KP);             // [C_NW][128] = the 24 KB of slabs
  double* wacc = reinterpret_cast<double*>(Xt + 64 * KP + C_NW * 1024);   // [2][64 * nslices]
  if (stat_slab)
    for (int i = threadIdx.x; i < 128 * nslices; i += C_THR) wacc[i] = 0.0;
  // unit order: the slices of one graph sit on workgroups of the same XCD (blockIdx % 8) and run
  // at the same time, so the graph's operator lists come from HBM once and from that XCD's L2
  // for the other slices (consecutive workgroups are on DIFFERENT XCDs: PMC showed 215 MB per
  // launch against 117 MB of compulsory bytes with the plain u = blockIdx order)
  const bool xcd_map = (gridDim.x % 8 == 0) && ((gridDim.x / 8) % sub == 0);
  for (int it = 0;; ++it) {
    int u;
    if (xcd_map) {
      const int per_xcd = gridDim.x / 8;                           // workgroups per XCD
      const int x = blockIdx.x % 8, i = blockIdx.x / 8;
      const int gg = it * (gridDim.x / sub) + x * (per_xcd / sub) + i / sub;
      u = gg * sub + i % sub;
      if (it * (int)gridDim.x >= units) break;
      if (gg >= B) continue;
    } else {
      u = it * gridDim.x + blockIdx.x;
      if (u >= units) break;
    }
    const int g = u / sub, rem = u - g * sub;
    const int slice = rem % nslices, part = rem / nslices;
    const int base = gptr[g], n = gptr[g + 1] - base;
    const int ksteps = (n + 15) >> 4;
    const int nrbn = (n + 31) >> 5, rb_per = (nrbn + rparts - 1) / rparts;
    const int rb_lo = part * rb_per, rb_hi = min(nrbn, rb_lo + rb_per);

    // ---- transpose the [n x 64] slice of X into LDS; k in [n, 16*ksteps) is zero-filled
    __syncthreads();
    if (!BNB) {
      const int piece = threadIdx.x & 7;
      constexpr int SU = 4;                         // rows in flight per thread
      for (int k0 = threadIdx.x >> 3; k0 < 16 * ksteps; k0 += SU * (C_THR / 8)) {
        uint4 v[SU];
#pragma unroll
        for (int q = 0; q < SU; ++q) {
          const int k = k0 + q * (C_THR / 8);
          v[q] = make_uint4(0u, 0u, 0u, 0u);
          if (k < n) v[q] = *reinterpret_cast<const uint4*>(X + (int64_t)(base + k) * ldx + 64 * slice + 8 * piece);
        }
#pragma unroll
        for (int q = 0; q < SU; ++q) {
          const int k = k0 + q * (C_THR / 8);
          if (k < 16 * ksteps) {
            const __half* hv = reinterpret_cast<const __half*>(&v[q]);
#pragma unroll
            for (int i = 0; i < 8; ++i) {          // column c = 8*piece + i lives in LDS row c/2 + 32*(c%2)
              const int c = 8 * piece + i;
              Xt[((c >> 1) + 32 * (c & 1)) * KP + k] = hv[i];
            }
          }
        }
      }
    } else {
      // the same transposition of dY, formed here from dX' (or dP), Yl, the keep bytes and coefficients
      const int piece = threadIdx.x & 7;
      const int F = 64 * nslices, c0 = 64 * slice + 8 * piece;
      float ca[8], cb[8], cm[8], ci[8], c1[8], c2[8], gp[8], cs[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        ca[i] = bb.coef[c0 + i]; cb[i] = bb.coef[F + c0 + i];
        cm[i] = bb.coef[2 * F + c0 + i]; ci[i] = bb.coef[3 * F + c0 + i];
        c1[i] = bb.bwc[c0 + i]; c2[i] = bb.bwc[F + c0 + i];
        cs[i] = 0.f;
        gp[i] = 0.f;
      }
      if (bb.dP) {
        const float inv = 1.0f / ((float)n + 1e-8f);
#pragma unroll
        for (int i = 0; i < 8; ++i) gp[i] = bb.dP[(int64_t)g * F + c0 + i] * inv;
      }
      constexpr int SU = 3;
      for (int k0 = threadIdx.x >> 3; k0 < 16 * ksteps; k0 += SU * (C_THR / 8)) {
        uint4 vy[SU], vx[SU];
        uint32_t kb[SU];
#pragma unroll
        for (int q = 0; q < SU; ++q) {
          const int k = k0 + q * (C_THR / 8);
          vy[q] = vx[q] = make_uint4(0u, 0u, 0u, 0u);
          kb[q] = 0xFFu;
          if (k < n) {
            vy[q] = *reinterpret_cast<const uint4*>(bb.Yl + (int64_t)(base + k) * bb.ldyl + c0);
            if (bb.dX) vx[q] = *reinterpret_cast<const uint4*>(bb.dX + (int64_t)(base + k) * ldx + c0);
            if (bb.mask) {
              const uint8_t* mp = bb.mask + (int64_t)(base + k) * (F >> 2) + (c0 >> 2);
              kb[q] = (uint32_t)(mp[0] & 0xFu) | ((uint32_t)(mp[1] & 0xFu) << 4);
            }
          }
        }
#pragma unroll
        for (int q = 0; q < SU; ++q) {
          const int k = k0 + q * (C_THR / 8);
          if (k < 16 * ksteps) {
            const __half* hy = reinterpret_cast<const __half*>(&vy[q]);
            const __half* hx = reinterpret_cast<const __half*>(&vx[q]);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
              float d = 0.f;
              if (k < n) {
                const float y = __half2float(hy[i]);
                const float gx = bb.dX ? __half2float(hx[i]) : gp[i];
                const float z = fmaf(ca[i], y, cb[i]);
                const float f = ((!bb.relu || z > 0.f) && ((kb[q] >> i) & 1u)) ? bb.scale : 0.f;
                d = bn_bwd_dy(ca[i], gx, f, c1[i], y, cm[i], ci[i], c2[i]);
                cs[i] += d;
              }
              const int c = 8 * piece + i;
              Xt[((c >> 1) + 32 * (c & 1)) * KP + k] = __float2half_rn(d);
            }
          }
        }
      }
      // db: column sums of dY over the graph.  Lanes with the same piece sit 8 apart in a wave; the 12
      // waves' partials meet in the (unused here) statistics area behind the slabs.
      if (part == 0) {
        float* csr = reinterpret_cast<float*>(Xt + 64 * KP + C_NW * 1024);          // [C_NW][64]
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          float v = cs[i];
          v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
          if (lane < 8) csr[wave * 64 + 8 * piece + i] = v;
        }
      }
    }
    __syncthreads();
    if (BNB && part == 0 && threadIdx.x < 64) {
      const float* csr = reinterpret_cast<const float*>(Xt + 64 * KP + C_NW * 1024);
      double tot = 0.0;
#pragma unroll
      for (int w2 = 0; w2 < C_NW; ++w2) tot += (double)csr[w2 * 64 + threadIdx.x];
      bb.cs_slab[(int64_t)g * 64 * nslices + 64 * slice + threadIdx.x] = tot;
    }

    double st1[2] = {0.0, 0.0}, st2[2] = {0.0, 0.0};
    for (int rb = rb_lo + wave; rb < rb_hi; rb += C_NW) {
      const int64_t row = (int64_t)g * NRB + rb;
      const __half* b0 = Xt + r * KP + 8 * h;       // column 2r   (LDS row c/2 + 32*(c%2))
      const __half* b1 = b0 + 32 * KP;              // column 2r+1
      f32x16 acc0 = {0}, acc1 = {0};

      // ---- dense list: operands straight from memory (1 KB contiguous per fragment)
      {
        const uint32_t d0 = doff[row];
        const int nd = (int)(doff[row + 1] - d0);
        if (nd > 0) {
          const int sv = lane < nd ? dstep[d0 + lane] : 0;         // lane i: k-step of item i (nd <= 64)
          const __half* ap = dfrag + ((int64_t)d0 * 64 + lane) * 8;
          h8 a[C_DAHEAD];
#pragma unroll
          for (int p = 0; p < C_DAHEAD; ++p) a[p] = *reinterpret_cast<const h8*>(ap + 512 * (int64_t)min(p, nd - 1));
          for (int i0 = 0; i0 < nd; i0 += C_DAHEAD) {
#pragma unroll
            for (int p = 0; p < C_DAHEAD; ++p) {
              h8 av = a[p];
              a[p] = *reinterpret_cast<const h8*>(ap + 512 * (int64_t)min(i0 + p + C_DAHEAD, nd - 1));
              const int it = min(i0 + p, nd - 1);
              if (i0 + p >= nd) av = h8{0};                          // past the list: adds nothing
              const int s = __builtin_amdgcn_readlane(sv, it);
              const h8 bv0 = *reinterpret_cast<const h8*>(b0 + 16 * s);
              const h8 bv1 = *reinterpret_cast<const h8*>(b1 + 16 * s);
              acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv0, acc0, 0, 0, 0);
              acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv1, acc1, 0, 0, 0);
            }
          }
        }
      }

      // ---- sparse list: scatter one chunk into a zero slab, read it as an operand, clear it again.
      // Two slabs alternate and the MFMAs of an item run one item later, on operands already read:
      // LDS executes a wave's accesses in order, so store -> load -> store needs only program order.
      {
        const uint32_t s0 = soff[row];
        const int ns4 = (int)(soff[row + 1] - s0) >> 2;              // (a multiple of four chunks)
        if (ns4 > 0) {
          const int sv = lane < 4 * ns4 ? sstep[s0 + lane] : 0;     // lane i: k-step of chunk i (<= 64)
          const uint4* ep4 = reinterpret_cast<const uint4*>(sent) + (int64_t)(s0 >> 2) * 64 + lane;
          uint4 ring[C_SAHEAD];
#pragma unroll
          for (int p = 0; p < C_SAHEAD; ++p) ring[p] = ep4[64 * (int64_t)min(p, ns4 - 1)];
          h8 av_p = {0}, b0_p = {0}, b1_p = {0};
          for (int c4 = 0; c4 < ns4; c4 += C_SAHEAD) {
#pragma unroll
            for (int p = 0; p < C_SAHEAD; ++p) {
              const uint4 e4 = ring[p];
              ring[p] = ep4[64 * (int64_t)min(c4 + p + C_SAHEAD, ns4 - 1)];
              const bool live = c4 + p < ns4;
              const int cb = 4 * min(c4 + p, ns4 - 1);
              const uint32_t ev[4] = {e4.x, e4.y, e4.z, e4.w};
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                unsigned short* slab = reinterpret_cast<unsigned short*>(frag2 + 512 * (q & 1));
                const uint32_t e = ev[q];
                const bool put = live && (e & 0xFFFFu) != C_PAD;
                if (put) slab[e & 0xFFFFu] = (unsigned short)(e >> 16);
                __builtin_amdgcn_wave_barrier();
                const int s = __builtin_amdgcn_readlane(sv, cb + q);
                h8 av = *reinterpret_cast<const h8*>(frag2 + 512 * (q & 1) + 8 * lane);
                const h8 bv0 = *reinterpret_cast<const h8*>(b0 + 16 * s);
                const h8 bv1 = *reinterpret_cast<const h8*>(b1 + 16 * s);
                __builtin_amdgcn_wave_barrier();
                if (put) slab[e & 0xFFFFu] = 0;
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(av_p, b0_p, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(av_p, b1_p, acc1, 0, 0, 0);
                av_p = av; b0_p = bv0; b1_p = bv1;                 // (a dead group read an all-zero slab)
              }
            }
          }
          acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(av_p, b0_p, acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(av_p, b1_p, acc1, 0, 0, 0);
        }
      }

      float2 bia = make_float2(0.f, 0.f);
      if (bias) bia = *reinterpret_cast<const float2*>(bias + 64 * slice + 2 * r);
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int orow = 32 * rb + (q & 3) + 8 * (q >> 2) + 4 * h;
        if (orow < n) {
          const __half2 hv = __floats2half2_rn(acc0[q] + bia.x, acc1[q] + bia.y);
          *reinterpret_cast<__half2*>(Y + (int64_t)(base + orow) * ldy + 64 * slice + 2 * r) = hv;
          if (stat_slab) {
            const float2 fv = __half22float2(hv);
            st1[0] += fv.x; st1[1] += fv.y;
            st2[0] += (double)fv.x * fv.x; st2[1] += (double)fv.y * fv.y;
          }
        }
      }
    }
    if (stat_slab) {
      __syncthreads();                               // every wave is done with its slabs
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        st1[t] += __shfl_xor(st1[t], 32, 64);
        st2[t] += __shfl_xor(st2[t], 32, 64);
        if (h == 0) { red[wave * 128 + 2 * r + t] = st1[t]; red[wave * 128 + 64 + 2 * r + t] = st2[t]; }
      }
      __syncthreads();
      if (threadIdx.x < 128) {
        double tot = 0.0;
#pragma unroll
        for (int w2 = 0; w2 < C_NW; ++w2) tot += red[w2 * 128 + threadIdx.x];
        wacc[(threadIdx.x >> 6) * 64 * nslices + 64 * slice + (threadIdx.x & 63)] += tot;
      }
      __syncthreads();
      *reinterpret_cast<uint4*>(frag2 + 8 * lane) = make_uint4(0u, 0u, 0u, 0u);        // slabs zero again
      *reinterpret_cast<uint4*>(frag2 + 512 + 8 * lane) = make_uint4(0u, 0u, 0u, 0u);
    }
  }
  if (stat_slab) {
    __syncthreads();
    for (int i = threadIdx.x; i < 128 * nslices; i += C_THR) stat_slab[(int64_t)blockIdx.x * 128 * nslices + i] = wacc[i];
  }
}

// transposed slice + the waves' fragment slabs + the statistics area (running column sums of the
// forward statistics, or the waves' bias-gradient partials of the BNB form: C_NW x 64 floats)
size_t c_lds(int P, int F) {
  const size_t area = (size_t)2 * F * sizeof(double) > (size_t)C_NW * 64 * sizeof(float) ? (size_t)2 * F * sizeof(double)
                                                                                       : (size_t)C_NW * 64 * sizeof(float);
  return ((size_t)64 * c_kp(P) + (size_t)C_NW * 1024) * sizeof(__half) + area;
}

bool pack_attr() {
  static bool done[CGNN_MAX_DEVICES] = {};
  bool& d = done[cgnn_device_ordinal()];
  if (!d) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_dense_pack<false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 32 * C_MAXP * 4) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(k_dense_pack<true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 32 * C_MAXP * 4) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(k_dense_agg_c<false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(k_dense_agg_c<true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return false;
    d = true;
  }
  return true;
}

}  // namespace

extern "C" {

int cgnn_dense_pack_count(const int32_t* rowptr, const int32_t* col, const float* coef,
                          const float* selfc, const int32_t* gptr, int32_t num_graphs, int32_t P,
                          uint32_t* counts, void* stream) {
  if (num_graphs < 0 || P <= 0 || P > C_MAXP || P % 64) return P > C_MAXP ? CGNN_EUNSUPPORTED : CGNN_EINVAL;
  if (num_graphs == 0) return CGNN_OK;
  if (num_graphs > 65535) return CGNN_EUNSUPPORTED;
  if (!rowptr || !col || !coef || !gptr || !counts) return CGNN_EINVAL;
  if (!pack_attr()) return CGNN_ELAUNCH;
  k_dense_pack<false><<<dim3((unsigned)(P >> 5), (unsigned)num_graphs), 256, (size_t)32 * P * sizeof(float),
                        cgnn_stream(stream)>>>(rowptr, col, coef, selfc, gptr, P, counts, nullptr, nullptr,
                                               nullptr, nullptr, nullptr);
  CGNN_CHECK_LAUNCH();
  return CGNN_OK;
}

int cgnn_dense_pack_fill(const int32_t* rowptr, const int32_t* col, const float* coef,
                         const float* selfc, const int32_t* gptr, int32_t num_graphs, int32_t P,
                         const uint32_t* fpos, void* dfrag, int32_t* dstep, uint32_t* sent,
                         int32_t* sstep, void* stream) {
  if (num_graphs < 0 || P <= 0 || P > C_MAXP || P % 64) return P > C_MAXP ? CGNN_EUNSUPPORTED : CGNN_EINVAL;
  if (num_graphs == 0) return CGNN_OK;
  if (num_graphs > 65535) return CGNN_EUNSUPPORTED;
  if (!rowptr || !col || !coef || !gptr || !fpos || !dfrag || !dstep || !sent || !sstep) return CGNN_EINVAL;
  if (!pack_attr()) return CGNN_ELAUNCH;
  k_dense_pack<true><<<dim3((unsigned)(P >> 5), (unsigned)num_graphs), 256, (size_t)32 * P * sizeof(float),
                       cgnn_stream(stream)>>>(rowptr, col, coef, selfc, gptr, P, nullptr, fpos,
                                              static_cast<__half*>(dfrag), dstep, sent, sstep);
  CGNN_CHECK_LAUNCH();
  return CGNN_OK;
}

int cgnn_dense_aggregate_c16(const void* dfrag, const int32_t* dstep, const uint32_t* doff,
                             const uint32_t* sent, const int32_t* sstep, const uint32_t* soff,
                             int32_t P, const int32_t* gptr, int32_t num_graphs, const void* X,
                             int64_t ldx, int32_t F, const float* bias, void* Y, int64_t ldy,
                             double* stat_slab, int64_t stat_slab_bytes, void* stream) {
  if (num_graphs < 0 || P <= 0 || F <= 0 || ldx < F || ldy < F) return CGNN_EINVAL;
  if (P > C_MAXP || P % 64 || F % 64 || ldx % 8 || c_lds(P, F) > 160 * 1024) return CGNN_EUNSUPPORTED;
  if (reinterpret_cast<uintptr_t>(X) & 15) return CGNN_EUNSUPPORTED;
  if (num_graphs == 0) return CGNN_OK;
  if (!dfrag || !dstep || !doff || !sent || !sstep || !soff || !gptr || !X || !Y) return CGNN_EINVAL;
  CGNN_NEED_BYTES(stat_slab, stat_slab_bytes, (int64_t)cgnn_fused_grid() * 2 * F * (int64_t)sizeof(double));
  if (!pack_attr()) return CGNN_ELAUNCH;
  const int grid = cgnn_fused_grid();
  int rparts = 1;
  while (rparts < 4 && (int64_t)num_graphs * (F / 64) * rparts * 2 <= grid) rparts *= 2;
  k_dense_agg_c<false><<<grid, C_THR, c_lds(P, F), cgnn_stream(stream)>>>(
      static_cast<const __half*>(dfrag), dstep, doff, sent, sstep, soff, P, gptr, num_graphs,
      static_cast<const __half*>(X), ldx, F / 64, bias, static_cast<__half*>(Y), ldy, stat_slab, rparts,
      DenseBnBwd{});
  CGNN_CHECK_LAUNCH();
  return CGNN_OK;
}

int cgnn_dense_aggregate_c16_bnbwd(const void* dfrag, const int32_t* dstep, const uint32_t* doff,
                                   const uint32_t* sent, const int32_t* sstep, const uint32_t* soff,
                                   int32_t P, const int32_t* gptr, int32_t num_graphs, const void* dX,
                                   int64_t lddx, const float* dP, const void* Yl, int64_t ldyl,
                                   const uint8_t* mask, const float* coef, const float* bwc, int32_t relu,
                                   float p_drop, int32_t F, void* dT, int64_t lddt, double* cs_slab, int64_t cs_slab_bytes,
                                   void* stream) {
  if (num_graphs < 0 || P <= 0 || F <= 0 || ldyl < F || lddt < F || (dX && lddx < F)) return CGNN_EINVAL;
  if ((!dX) == (!dP) || !Yl || !coef || !bwc || !cs_slab || p_drop < 0.f || p_drop >= 1.f || (p_drop > 0.f && !mask))
    return CGNN_EINVAL;
  if (P > C_MAXP || P % 64 || F % 64 || ldyl % 8 || (dX && lddx % 8) || c_lds(P, F) > 160 * 1024) return CGNN_EUNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(dX) | reinterpret_cast<uintptr_t>(Yl)) & 15) return CGNN_EUNSUPPORTED;
  if (num_graphs == 0) return CGNN_OK;
  if (!dfrag || !dstep || !doff || !sent || !sstep || !soff || !gptr || !dT) return CGNN_EINVAL;
  CGNN_NEED_BYTES(cs_slab, cs_slab_bytes, (int64_t)num_graphs * F * (int64_t)sizeof(double));
  if (!pack_attr()) return CGNN_ELAUNCH;
  const int grid = cgnn_fused_grid();
  int rparts = 1;
  while (rparts < 4 && (int64_t)num_graphs * (F / 64) * rparts * 2 <= grid) rparts *= 2;
  DenseBnBwd bb;
  bb.dX = static_cast<const __half*>(dX);
  bb.dP = dP;
  bb.Yl = static_cast<const __half*>(Yl);
  bb.ldyl = ldyl;
  bb.mask = p_drop > 0.f ? mask : nullptr;
  bb.coef = coef;
  bb.bwc = bwc;
  bb.relu = relu;
  bb.scale = p_drop > 0.f ? (float)(1.0 / (1.0 - (double)p_drop)) : 1.0f;
  bb.cs_slab = cs_slab;
  k_dense_agg_c<true><<<grid, C_THR, c_lds(P, F), cgnn_stream(stream)>>>(
      static_cast<const __half*>(dfrag), dstep, doff, sent, sstep, soff, P, gptr, num_graphs,
      static_cast<const __half*>(dX), dX ? lddx : 0, F / 64, nullptr, static_cast<__half*>(dT), lddt, nullptr,
      rparts, bb);
  CGNN_CHECK_LAUNCH();
  return CGNN_OK;
}

}  // extern "C"
