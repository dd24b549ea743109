// band_aggregate.hip -- the dense part of the aggregation operator of LARGE graphs in fp32, on the bf16
// matrix pipe with exactly split operands (gfx950).
//
//   Y[r,:] = ( sum over the edges of row r that fall into DENSE fragments  c_e * X[col_e,:] ) (/ rowdiv[r])
//
// Replaces, for graphs of more than 384 nodes (BASELINE config 5's 1000-ROI parcellation in the
// reference's own arithmetic), most of the gather -> mul -> scatter_add_ of models.py:112-114 / :146-149
// and of their autograd transposes.  The gather kernel (aggregate.hip) reads ~100 neighbour rows of 1 KB
// per output row out of L2 and is bound by exactly that traffic (6.4 GB per [64000 x 256] launch,
// 225-235 us).  Small-world connectomes are bimodal (dense_aggregate_c16.hip): the 32-row x 16-source
// fragments on the lattice band are nearly full and hold ~90 % of the edges.  Those fragments are
// applied as matrix products here -- fp32 exact to rounding: both operands are cut into three bf16
// pieces by truncation (split_bf16.h) and the six partial products of order <= 2 are accumulated in fp32
// by v_mfma_f32_32x32x16_bf16 -- and the gather kernel (cgnn_aggregate_acc_f32, launched next) adds the
// ~15 % of edges outside them (a CSR filtered by the caller), the self-loop term and the bias.
//
//   cgnn_band_pack_f32       the dense fragments of one CSR ordering as MFMA A operands, already split:
//                            bfrag[item][piece h|m|l][lane][8 bf16] (3 KB per fragment), bstep[item] = its
//                            k-step; items of a (graph, row block) are contiguous (boff).  Static per batch.
//   cgnn_band_aggregate_f32  one wave per (graph, 32-row block, 64-column panel): streams the block's
//                            fragments (A pieces: three 16-byte loads per lane; B: the 16 source rows'
//                            64 columns straight from X, split in registers), twelve MFMAs each, and
//                            writes the [32 x 64] result to Y.  No LDS: the operands are register-shaped as
//                            they arrive, and the four waves of a workgroup (four panels of one row
//                            block) share the A pieces through L1.
#include "common.h"
#include "split_bf16.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BD_MAXP = 1024;
constexpr uint32_t BD_NONE = 0xFFFFFFFFu;      // fpos of a fragment that is not in the dense list

// ------------------------------------------------------------------------------------ builder
// one block per (row block, graph): the 32 rows accumulate in LDS (serial per row, COO order:
// reproducible; duplicate edges add up in fp32), then every listed k-step's 512 values are split and
// written in operand order: lane ln = slot >> 3 holds A[i = ln & 31][k = 16 s + 8 (ln >> 5) + e]
__global__ void __launch_bounds__(256) k_band_pack(
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, const float* __restrict__ coef,
    const int32_t* __restrict__ gptr, int P, const uint32_t* __restrict__ fpos,
    uint4* __restrict__ bfrag, int32_t* __restrict__ bstep) {
  extern __shared__ float rows[];                 // [32][P]
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
    }
  }
  __syncthreads();
  const int64_t frag0 = ((int64_t)g * NRB + rb) * S;
  for (int s = 0; s < S; ++s) {
    const uint32_t fp = fpos[frag0 + s];              // block-uniform
    if (fp == BD_NONE) continue;
    if (t < 64) {
      const int i = t & 31, k0 = 16 * s + 8 * (t >> 5);
      const float* src = rows + i * P + k0;
      const Split8 sp = split8(make_float4(src[0], src[1], src[2], src[3]), make_float4(src[4], src[5], src[6], src[7]));
      uint4* out = bfrag + (int64_t)fp * 192 + t;     // [piece][lane]
      out[0] = sp.h;
      out[64] = sp.m;
      out[128] = sp.l;
      if (t == 0) bstep[fp] = s;
    }
  }
}

// ----------------------------------------------------------------------------------- aggregate
__device__ __forceinline__ f32x16 bd_mfma(const uint4& a, const uint4& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

struct BdItem { uint4 ah, am, al; float2 x[8]; };   // one fragment's operands as they arrive

// one wave per (graph, 32-row block, 64-column panel); lane (j, h) owns the panel's columns 2j and 2j+1:
// one 8-byte load per source row feeds the B operands of two [32 x 32] products (even / odd columns),
// which share the A pieces and run as two independent MFMA chains.  Workgroup = four panels of one row
// block.  blockIdx -> task is XCD-aware: consecutive workgroups go to different XCDs, so the task list is
// cut into eight contiguous runs -- a graph's feature rows and fragments stay in ONE L2.
// (Measured and not kept: the remaining edges walked by the same wave on the same accumulators, Y
// written once -- 202 us against 66 + 58 us for the two launches: the walk is 160 dependent-latency
// 8-byte gathers per wave where the gather kernel issues 16-byte loads, a whole row per wave.  And this
// kernel adding onto the gather kernel's Y: its read-modify-write epilogue was 16 of 57 us, exposed at
// the end of every wave; the gather kernel hides the same read among its row's gathers.  And the loads
// of fragment i + 2 issued before fragment i's products (three fragments' operands live, 132 registers,
// three waves per SIMD instead of four): 49.9 us against 47.4 back to back.)
__global__ void __launch_bounds__(256) k_band_agg(
    const uint4* __restrict__ bfrag, const int32_t* __restrict__ bstep, const int32_t* __restrict__ boff,
    int P, const int32_t* __restrict__ gptr, int B, const float* __restrict__ X, int64_t ldx, int F,
    const float* __restrict__ rowdiv, const float* __restrict__ Yadd, int64_t ldadd, float* __restrict__ Y,
    int64_t ldy) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 31, h = lane >> 5;
  const int NRB = P >> 5;
  const int panels = (F + 63) >> 6, pgroups = (panels + 3) >> 2;   // workgroups per (graph, row block)
  const int64_t tasks = (int64_t)B * NRB * pgroups;
  const int64_t per_xcd = gridDim.x >> 3;                          // (the grid is a multiple of 8)
  for (int64_t task = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3); task < tasks; task += gridDim.x) {
    const int pg = (int)(task % pgroups);
    const int64_t row = task / pgroups;                     // = g * NRB + rb
    const int g = (int)(row / NRB), rb = (int)(row - (int64_t)g * NRB);
    const int panel = 4 * pg + wave;
    const int base = gptr[g], n = gptr[g + 1] - base;
    if (panel >= panels || 32 * rb >= n) continue;          // (wave-uniform)
    const int d0 = boff[row], nd = boff[row + 1] - d0;
    const int c0 = 64 * panel + 2 * j;
    const bool colok = c0 < F;                              // F is even: 2j+1 is inside with 2j
    const float* xcol = X + (colok ? c0 : 0);
    // the k-steps of the row block's fragments: one load up front (a row block has at most P/16 <= 64),
    // so the source-row addresses of a fragment do not wait on a load of their own
    const int steps_v = lane < nd ? bstep[d0 + lane] : 0;
    auto fetch = [&](BdItem& it, int item) __attribute__((always_inline)) {
      const uint4* ap = bfrag + (int64_t)(d0 + item) * 192 + lane;
      it.ah = ap[0];
      it.am = ap[64];
      it.al = ap[128];
      const int k0 = 16 * __builtin_amdgcn_readlane(steps_v, item) + 8 * h;   // B[k = 8 h + e][col]
#pragma unroll
      for (int e = 0; e < 8; ++e)
        it.x[e] = (k0 + e < n && colok) ? *reinterpret_cast<const float2*>(xcol + (int64_t)(base + k0 + e) * ldx)
                                        : make_float2(0.f, 0.f);
    };
    f32x16 acc0 = {0}, acc1 = {0};
    if (nd > 0) {
      BdItem cur, nxt;
      fetch(cur, 0);
      nxt = cur;
      for (int item = 0; item < nd; ++item) {
        if (item + 1 < nd) fetch(nxt, item + 1);
        const Split8 b0 = split8(make_float4(cur.x[0].x, cur.x[1].x, cur.x[2].x, cur.x[3].x),
                                 make_float4(cur.x[4].x, cur.x[5].x, cur.x[6].x, cur.x[7].x));
        const Split8 b1 = split8(make_float4(cur.x[0].y, cur.x[1].y, cur.x[2].y, cur.x[3].y),
                                 make_float4(cur.x[4].y, cur.x[5].y, cur.x[6].y, cur.x[7].y));
        acc0 = bd_mfma(cur.al, b0.h, acc0);                 // smallest terms first
        acc1 = bd_mfma(cur.al, b1.h, acc1);
        acc0 = bd_mfma(cur.ah, b0.l, acc0);
        acc1 = bd_mfma(cur.ah, b1.l, acc1);
        acc0 = bd_mfma(cur.am, b0.m, acc0);
        acc1 = bd_mfma(cur.am, b1.m, acc1);
        acc0 = bd_mfma(cur.am, b0.h, acc0);
        acc1 = bd_mfma(cur.am, b1.h, acc1);
        acc0 = bd_mfma(cur.ah, b0.m, acc0);
        acc1 = bd_mfma(cur.ah, b1.m, acc1);
        acc0 = bd_mfma(cur.ah, b0.h, acc0);
        acc1 = bd_mfma(cur.ah, b1.h, acc1);
        cur = nxt;
      }
    }
    if (!colok) continue;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int orow = 32 * rb + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (orow < n) {
        float v0 = acc0[r], v1 = acc1[r];
        if (rowdiv) {
          const float d = rowdiv[base + orow];
          v0 = v0 / d;
          v1 = v1 / d;
        }
        if (Yadd) {                                         // GraphSAGE backward: dX = dX1 + A^T (dA / den)
          const float2 ya = *reinterpret_cast<const float2*>(Yadd + (int64_t)(base + orow) * ldadd + c0);
          v0 += ya.x;
          v1 += ya.y;
        }
        *reinterpret_cast<float2*>(Y + (int64_t)(base + orow) * ldy + c0) = make_float2(v0, v1);
      }
    }
  }
}

}  // namespace

extern "C" {

int cgnn_band_pack_f32(const int32_t* rowptr, const int32_t* col, const float* coef, const int32_t* gptr,
                       int32_t num_graphs, int32_t P, const uint32_t* fpos, void* bfrag, int32_t* bstep,
                       void* stream) {
  if (num_graphs < 0 || P <= 0 || P > BD_MAXP || P % 64) return P > BD_MAXP ? CGNN_EUNSUPPORTED : CGNN_EINVAL;
  if (num_graphs == 0) return CGNN_OK;
  if (num_graphs > 65535) return CGNN_EUNSUPPORTED;
  if (!rowptr || !col || !coef || !gptr || !fpos || !bfrag || !bstep) return CGNN_EINVAL;
  static bool attr[CGNN_MAX_DEVICES] = {};
  bool& done = attr[cgnn_device_ordinal()];
  if (!done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_band_pack), hipFuncAttributeMaxDynamicSharedMemorySize,
                            32 * BD_MAXP * 4) != hipSuccess)
      return CGNN_ELAUNCH;
    done = true;
  }
  k_band_pack<<<dim3((unsigned)(P >> 5), (unsigned)num_graphs), 256, (size_t)32 * P * sizeof(float),
                cgnn_stream(stream)>>>(rowptr, col, coef, gptr, P, fpos, static_cast<uint4*>(bfrag), bstep);
  CGNN_CHECK_LAUNCH();
  return CGNN_OK;
}

int cgnn_band_aggregate_f32(const void* bfrag, const int32_t* bstep, const int32_t* boff, int32_t P,
                            const int32_t* gptr, int32_t num_graphs, const float* X, int64_t ldx, int32_t F,
                            const float* rowdiv, const float* Yadd, int64_t ldadd, float* Y, int64_t ldy,
                            void* stream) {
  if (num_graphs < 0 || P <= 0 || F <= 0 || ldx < F || ldy < F || (Yadd && ldadd < F)) return CGNN_EINVAL;
  if (P > BD_MAXP || P % 64 || F % 32 || ldx % 2 || ldy % 2 || (reinterpret_cast<uintptr_t>(X) & 7) ||
      (reinterpret_cast<uintptr_t>(Y) & 7) || (Yadd && (ldadd % 2 || (reinterpret_cast<uintptr_t>(Yadd) & 7))))
    return CGNN_EUNSUPPORTED;                               // (8-byte accesses of column pairs)
  if (num_graphs == 0) return CGNN_OK;
  if (!bfrag || !bstep || !boff || !gptr || !X || !Y || X == Y) return CGNN_EINVAL;
  const int64_t tasks = (int64_t)num_graphs * (P >> 5) * (((F + 63) / 64 + 3) / 4);
  const unsigned grid = (unsigned)(((tasks < 65536 ? tasks : 65536) + 7) / 8 * 8);
  k_band_agg<<<grid, 256, 0, cgnn_stream(stream)>>>(static_cast<const uint4*>(bfrag), bstep, boff, P, gptr, num_graphs,
                                                    X, ldx, F, rowdiv, Yadd, ldadd, Y, ldy);
  CGNN_CHECK_LAUNCH();
  return CGNN_OK;
}

}  // extern "C"
