// gemm_h16.hip -- projection kernels of the fp16-STORAGE GCN path (BASELINE config 5: 1000-ROI
// graphs, hidden 256; gcn_half_path.py).  Arithmetic of models.py:111 (T = X W^T) and of its
// autograd backward (dX = dT W, dW = dT^T X) with activations stored as IEEE half, fp32
// accumulation on v_mfma_f32_32x32x16_f16, parameters and their gradients fp32.
//
//   k_hgemm<NT, BT>   forward / backward-input, weight-stationary: the whole weight (<= 256 x 256,
//                     converted fp32 -> half while it is laid out as MFMA B fragments) sits in LDS
//                     (128 KB) for the lifetime of a persistent 8-wave workgroup; every wave
//                     streams its own 32-row blocks of the activation with line-shaped 16-byte
//                     loads (4 lanes = one 64-byte half line of a row), turns a [32 x 32] chunk
//                     into the lane = row operand layout through a wave-private 2.5 KB LDS slab
//                     and runs 2 x NT MFMAs on it.  No workgroup barrier after the panel fill.  A
//                     lane ends up with NT CONSECUTIVE output columns of each of its 16 rows
//                     (tile t <-> columns {NT*j + t}), so a result row leaves as one 16-byte store.
//                     HBM-bound: 2 * M * 256 * 2 bytes per launch, 3.4 us of matrix pipe.
//   k_hgemm_wgrad<KT> dW = dY^T X: the reduction runs over the ROW index of both row-major
//                     operands, so both MFMA operands are columns of a staged tile: the
//                     workgroup stages [32 rows x 128 columns] of each operand in LDS row-major
//                     (as it arrives, 16-byte loads, 4 stages in flight in registers) and reads the
//                     operands with ds_read_b64_tr_b16 (the LDS transposing read of gfx950; 320-byte
//                     row stride = conflict-free).  A workgroup owns a [128 x 128] (or [128 x 64])
//                     tile of dW over one contiguous run of rows; the tiles of one run sit on
//                     the same XCD, so the operands' second read is an L2 hit.  fp32 partials
//                     [runs][N x K], folded in fixed order by k_hgemm_fold (fp64 accumulate).
//
// v_mfma_f32_32x32x16_f16 operand maps (cdna_hip_programming.md section 3):
//   A: lane l holds A[i = l&31][k = 8*(l>>5) + e]     B: lane l holds B[k = 8*(l>>5) + e][j = l&31]
//   C/D reg r of lane l: row = (r&3) + 8*(r>>2) + 4*(l>>5), col = l&31
#include <hip/hip_fp16.h>
#include "common.h"

int cgnn_fused_grid();

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef short s4 __attribute__((ext_vector_type(4)));
typedef short s8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int HG_NW = 8;                     // waves per workgroup (one workgroup per CU)
constexpr int HG_THR = HG_NW * 64;
constexpr int HG_SLD = 40;                   // halfs per row of a wave's [32 x 32] staging slab (80 B)
constexpr int HG_RING = 4;                   // activation chunks in flight per wave
constexpr int HG_MAX_PANEL = 131072;         // bytes of B fragments: K * NT * 64 <= this

__device__ __forceinline__ h8 as_h8(const uint4& v) { return __builtin_bit_cast(h8, v); }
struct Chunk { cgnn_u32x4 a, b; };               // one wave-load pair: rows lrow and 16 + lrow of a [32 x 32] chunk

// --------------------------------------------------------------- forward / backward input
// Y[M, 32*NT] (half) = X[M, K] (half) * B + bias, fp32 accumulate.
//   BT = false: B[k][c] = W[c*ldw + k] for k < Kw, else 0   (Y = X W^T, W fp32 [cols, Kw])
//   BT = true : B[k][c] = W[k*ldw + c]                       (dX = dY W,  W fp32 [K, cols])
//
// STATS (optional epilogue, per-workgroup fp64 partials -> stat_slab[blockIdx.x][2 * cols]):
//   1  BatchNorm forward statistics of the (half-rounded) output: sum y | sum y^2 -- the projection
//      in front of a BatchNorm leaves them behind instead of a statistics pass re-reading Y;
// (The mirror image for the backward pass -- the BatchNorm-backward sums of the layer below in the
// epilogue of dX = dT W, from that layer's output, keep bytes and coefficients -- was built and
// measured: 38 us against 21 + 14 for the product and a separate statistics pass.  With one 32-row
// block per wave at the config-5 shape nothing hides the epilogue's loads and 128 coefficient reads.)
struct HgStats {
  double* slab;                // [gridDim.x][2 * cols]
};

template <int NT, bool BT, int STATS>
__global__ void __launch_bounds__(HG_THR) k_hgemm(const __half* __restrict__ X, int64_t ldx, int K,
                                                  const float* __restrict__ W, int ldw, int Kw,
                                                  const float* __restrict__ bias,
                                                  __half* __restrict__ Y, int64_t ldy, int64_t M, HgStats hs) {
  extern __shared__ __attribute__((aligned(16))) unsigned char hg_lds[];
  uint4* Wl = reinterpret_cast<uint4*>(hg_lds);                       // [(s*NT + t)][lane] B fragments
  const int nsteps = K / 16, nchunks = K / 32;
  // (with STATS the panel region is at least HG_NW x 2 x cols doubles: the final fold reuses it)
  const size_t panel_bytes = STATS ? ((size_t)nsteps * NT * 64 * 16 > (size_t)HG_NW * 2 * 32 * NT * sizeof(double)
                                          ? (size_t)nsteps * NT * 64 * 16 : (size_t)HG_NW * 2 * 32 * NT * sizeof(double))
                                   : (size_t)nsteps * NT * 64 * 16;
  __half* stg = reinterpret_cast<__half*>(hg_lds + panel_bytes) + (threadIdx.x >> 6) * (32 * HG_SLD);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 31, h = lane >> 5;
  const int lrow = lane >> 2, lpc = lane & 3;
  const int64_t nrb = (M + 31) / 32;
  const int64_t stride = (int64_t)gridDim.x * HG_NW;

  // the activation stream starts before the panel is built: its latency hides behind the fill
  Chunk ring0, ring1, ring2, ring3;       // (named: an indexed array of these lands in scratch)
  auto request = [&](Chunk& buf, int64_t rb, int c) __attribute__((always_inline)) {
    int64_t row0 = rb * 32 + lrow, row1 = row0 + 16;
    if (row0 >= M) row0 = M - 1;                       // clamp: loads stay in bounds, rows unused
    if (row1 >= M) row1 = M - 1;
    buf.a = *reinterpret_cast<const cgnn_u32x4*>(X + row0 * ldx + 32 * c + 8 * lpc);
    buf.b = *reinterpret_cast<const cgnn_u32x4*>(X + row1 * ldx + 32 * c + 8 * lpc);
  };
  int64_t rb = (int64_t)blockIdx.x * HG_NW + wave;    // block being computed, its chunk
  int c = 0;
  int64_t qrb = rb;                                   // next chunk to request
  int qc = 0;
  auto advance = [&](int64_t& b, int& ch) { if (++ch == nchunks) { ch = 0; b += stride; } };
  if (qrb < nrb) { request(ring0, qrb, qc); advance(qrb, qc); }
  if (qrb < nrb) { request(ring1, qrb, qc); advance(qrb, qc); }
  if (qrb < nrb) { request(ring2, qrb, qc); advance(qrb, qc); }
  if (qrb < nrb) { request(ring3, qrb, qc); advance(qrb, qc); }

  // ---- the weight panel: fp32 -> half, MFMA-fragment-major
  if (BT) {
    for (int idx = threadIdx.x; idx < nsteps * 64; idx += HG_THR) {
      const int ln = idx & 63, s = idx >> 6, jj = ln & 31, hh = ln >> 5;
      float v[8][NT];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float* wp = W + (int64_t)(16 * s + 8 * hh + e) * ldw + NT * jj;
#pragma unroll
        for (int t = 0; t < NT; t += 2) {
          const float2 q = *reinterpret_cast<const float2*>(wp + t);
          v[e][t] = q.x; v[e][t + 1] = q.y;
        }
      }
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        h8 f;
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = (_Float16)v[e][t];
        Wl[(s * NT + t) * 64 + ln] = __builtin_bit_cast(uint4, f);
      }
    }
  } else {
    const bool vec = (ldw % 4 == 0) && Kw == K;
    for (int idx = threadIdx.x; idx < nsteps * NT * 64; idx += HG_THR) {
      const int ln = idx & 63, st = idx >> 6, jj = ln & 31, hh = ln >> 5;
      const int t = st % NT, s = st / NT;
      const float* wp = W + (int64_t)(NT * jj + t) * ldw + 16 * s + 8 * hh;
      h8 f;
      if (vec) {
        const float4 a = *reinterpret_cast<const float4*>(wp), b = *reinterpret_cast<const float4*>(wp + 4);
        f[0] = (_Float16)a.x; f[1] = (_Float16)a.y; f[2] = (_Float16)a.z; f[3] = (_Float16)a.w;
        f[4] = (_Float16)b.x; f[5] = (_Float16)b.y; f[6] = (_Float16)b.z; f[7] = (_Float16)b.w;
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = (16 * s + 8 * hh + e < Kw) ? (_Float16)wp[e] : (_Float16)0.f;
      }
      Wl[st * 64 + ln] = __builtin_bit_cast(uint4, f);
    }
  }
  __syncthreads();

  float bv[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) bv[t] = bias ? bias[NT * j + t] : 0.f;
  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = f32x16{0};
  double st1[STATS ? NT : 1], st2[STATS ? NT : 1];     // per-lane fp64 partial sums (sum y^2 cancels against mean^2)
#pragma unroll
  for (int t = 0; t < (STATS ? NT : 1); ++t) st1[t] = st2[t] = 0.0;
  auto process = [&](Chunk& buf) __attribute__((always_inline)) {
    *reinterpret_cast<cgnn_u32x4*>(stg + lrow * HG_SLD + 8 * lpc) = buf.a;
    *reinterpret_cast<cgnn_u32x4*>(stg + (16 + lrow) * HG_SLD + 8 * lpc) = buf.b;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const h8 a0 = as_h8(*reinterpret_cast<const uint4*>(stg + j * HG_SLD + 8 * h));
    const h8 a1 = as_h8(*reinterpret_cast<const uint4*>(stg + j * HG_SLD + 16 + 8 * h));
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (qrb < nrb) { request(buf, qrb, qc); advance(qrb, qc); }
    const uint4* wf = Wl + (size_t)(2 * c) * NT * 64 + lane;
#pragma unroll
    for (int t = 0; t < NT; ++t)
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, as_h8(wf[t * 64]), acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < NT; ++t)
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, as_h8(wf[(NT + t) * 64]), acc[t], 0, 0, 0);
    if (c + 1 == nchunks) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (row < M) {
          _Float16 o[NT];
#pragma unroll
          for (int t = 0; t < NT; ++t) o[t] = (_Float16)(acc[t][r] + bv[t]);
          if (STATS == 1) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
              const float v = (float)o[t];
              st1[t] += (double)v;
              st2[t] += (double)v * (double)v;
            }
          }
          __half* yp = Y + row * ldy + NT * j;
          if (NT == 8) {
            h8 v;
#pragma unroll
            for (int t = 0; t < 8; ++t) v[t] = o[t % NT];
            *reinterpret_cast<uint4*>(yp) = __builtin_bit_cast(uint4, v);
          } else if (NT == 4) {
            h4 v = {o[0], o[1], o[2 % NT], o[3 % NT]};
            *reinterpret_cast<uint2*>(yp) = __builtin_bit_cast(uint2, v);
          } else {
            h2 v = {o[0], o[1]};
            *reinterpret_cast<uint32_t*>(yp) = __builtin_bit_cast(uint32_t, v);
          }
        }
      }
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = f32x16{0};
    }
    advance(rb, c);
  };
  static_assert(HG_RING == 4, "the ring is unrolled by hand (constant register indices)");
  while (rb < nrb) {
    process(ring0);
    if (rb >= nrb) break;
    process(ring1);
    if (rb >= nrb) break;
    process(ring2);
    if (rb >= nrb) break;
    process(ring3);
  }
  if (STATS) {
    // lanes l and l + 32 hold the same columns; the waves' partials meet in the (now idle) panel
    constexpr int cols = 32 * NT;
    __syncthreads();
    double* red = reinterpret_cast<double*>(hg_lds);             // [HG_NW][2 * cols]
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const double a1 = (double)st1[t] + (double)__shfl_xor(st1[t], 32, 64);
      const double a2 = (double)st2[t] + (double)__shfl_xor(st2[t], 32, 64);
      if (h == 0) {
        red[wave * 2 * cols + NT * j + t] = a1;
        red[wave * 2 * cols + cols + NT * j + t] = a2;
      }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 2 * cols; e += HG_THR) {
      double tot = 0.0;
#pragma unroll
      for (int w2 = 0; w2 < HG_NW; ++w2) tot += red[w2 * 2 * cols + e];
      hs.slab[(int64_t)blockIdx.x * 2 * cols + e] = tot;
    }
  }
}

// ---------------------------------------------------------------------------- backward weight
constexpr int WG_RS = 160;                  // halfs per staged row: 128 + 32 pad = 320 B (see above)
constexpr int WG_ROWS = 32;                 // rows per stage
constexpr int WG_DEPTH = 4;                 // stages in flight in registers

__device__ __forceinline__ h8 tr_frag(const __half* img, int col0, int lane) {
  // the 32x32x16 operand whose reduction index runs down the ROWS of a row-major LDS image:
  // 16-lane group G reads the 4-row x 16-column blocks at rows 8h + {0..3}, 8h + {4..7}
  const int G = lane >> 4, q = (lane & 15) >> 2, p = lane & 3, hh = G >> 1;
  const __half* a = img + (8 * hh + q) * WG_RS + col0 + 16 * (G & 1) + 4 * p;
  typedef __attribute__((address_space(3))) s4 lds_s4;
  const s4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(a));
  const s4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(a + 4 * WG_RS));
  return __builtin_bit_cast(h8, __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7));
}

// partial[run][n][k] = sum_{m in run} dY[m][n] X[m][k] for this workgroup's [32*WN x 32*KT*WK] tile:
// WN x WK waves compute (n strips of 32, k strips of 32*KT); all 8 waves stage the operands.
template <int KT, int WN, int WK>
__global__ void __launch_bounds__(HG_THR) k_hgemm_wgrad(const __half* __restrict__ dY, int64_t lddy,
                                                        const __half* __restrict__ X, int64_t ldx,
                                                        float* __restrict__ slab, int64_t M, int N,
                                                        int K, int nruns, int64_t per) {
  static_assert(WN * WK <= HG_NW && 32 * WN <= 128 && 32 * KT * WK <= 128, "tile exceeds the staged image");
  constexpr int TNW = 32 * WN, TKW = 32 * KT * WK;                // n and k width of the workgroup tile
  __shared__ __attribute__((aligned(16))) __half img[2][2][WG_ROWS * WG_RS];   // [buffer][dY | X]
  const int ntk = K / TKW, ntiles = (N / TNW) * ntk;
  int run, tile;
  if (nruns % 8 == 0) {          // the tiles of one run on one XCD (blockIdx % 8): shared L2
    const int x = blockIdx.x & 7, y = blockIdx.x >> 3;
    run = x + 8 * (y / ntiles);
    tile = y % ntiles;
  } else {
    run = blockIdx.x / ntiles;
    tile = blockIdx.x % ntiles;
  }
  if (run >= nruns) return;                                        // (workgroup-uniform)
  const int n0 = TNW * (tile / ntk), k0 = TKW * (tile % ntk);
  const int64_t mbeg = min(M, (int64_t)run * per), mend = min(M, mbeg + per);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave % WN, wk = wave / WN;
  const bool computes = wave < WN * WK;                            // (wave-uniform)
  constexpr int AP = TNW / 8, XP = TKW / 8;                        // 16-byte pieces per staged row
  const int arow = tid / AP, apc = tid % AP;
  const bool aact = tid < WG_ROWS * AP;
  const int xrow = tid / XP, xpc = tid % XP;
  const bool xact = tid < WG_ROWS * XP;

  cgnn_u32x4 ra[WG_DEPTH], rx[WG_DEPTH];
  auto issue = [&](int slot, int64_t m0) __attribute__((always_inline)) {
    const cgnn_u32x4 z = {0u, 0u, 0u, 0u};
    ra[slot] = z; rx[slot] = z;
    if (aact && m0 + arow < mend)
      ra[slot] = *reinterpret_cast<const cgnn_u32x4*>(dY + (m0 + arow) * lddy + n0 + 8 * apc);
    if (xact && m0 + xrow < mend)
      rx[slot] = *reinterpret_cast<const cgnn_u32x4*>(X + (m0 + xrow) * ldx + k0 + 8 * xpc);
  };
#pragma unroll
  for (int d = 0; d < WG_DEPTH; ++d) issue(d, mbeg + (int64_t)WG_ROWS * d);

  f32x16 acc[KT];
#pragma unroll
  for (int t = 0; t < KT; ++t) acc[t] = f32x16{0};
  for (int64_t m0 = mbeg; m0 < mend; m0 += (int64_t)WG_ROWS * WG_DEPTH) {
#pragma unroll
    for (int d = 0; d < WG_DEPTH; ++d) {
      if (m0 + (int64_t)WG_ROWS * d >= mend) break;                // (uniform)
      __half* ia = img[d & 1][0];
      __half* ix = img[d & 1][1];
      if (aact) *reinterpret_cast<cgnn_u32x4*>(ia + arow * WG_RS + 8 * apc) = ra[d];
      if (xact) *reinterpret_cast<cgnn_u32x4*>(ix + xrow * WG_RS + 8 * xpc) = rx[d];
      issue(d, m0 + (int64_t)WG_ROWS * (d + WG_DEPTH));
      // one barrier per stage: a wave that writes buffer b again (two stages on) has passed the
      // barrier of the stage between, which every wave reaches only after computing on b
      __syncthreads();
      if (computes) {                 // whole waves: EXEC stays all ones for the transposing reads
#pragma unroll
        for (int ms = 0; ms < WG_ROWS / 16; ++ms) {
          const h8 a = tr_frag(ia + 16 * ms * WG_RS, 32 * wn, lane);
#pragma unroll
          for (int t = 0; t < KT; ++t) {
            const h8 b = tr_frag(ix + 16 * ms * WG_RS, 32 * KT * wk + 32 * t, lane);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[t], 0, 0, 0);
          }
        }
      }
    }
  }
  if (!computes) return;
  const int j = lane & 31, h = lane >> 5;
  float* out = slab + (int64_t)run * N * K;
#pragma unroll
  for (int t = 0; t < KT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = n0 + 32 * wn + (r & 3) + 8 * (r >> 2) + 4 * h;
      out[(int64_t)n * K + k0 + 32 * KT * wk + 32 * t + j] = acc[t][r];
    }
}

// dW[n*ldw + k] = sum_runs slab[run][n*K + k], k < Kw; fixed order, fp64 accumulate
__global__ void __launch_bounds__(256) k_hgemm_fold(const float* __restrict__ slab, int nruns, int N, int K,
                                                    int Kw, float* __restrict__ dW, int ldw) {
  const int i = blockIdx.x * 16 + (threadIdx.x & 15);              // element of [N x K]
  const int part = threadIdx.x >> 4;                               // 0..15
  const int64_t elems = (int64_t)N * K;
  double s = 0.0;
  if (i < elems)
    for (int r = part; r < nruns; r += 16) s += (double)slab[(int64_t)r * elems + i];
  __shared__ double red[256];
  red[threadIdx.x] = s;
  __syncthreads();
  if (part == 0 && i < elems) {
    double t = 0.0;
#pragma unroll
    for (int p2 = 0; p2 < 16; ++p2) t += red[p2 * 16 + (threadIdx.x & 15)];
    const int n = i / K, k = i % K;
    if (k < Kw) dW[(int64_t)n * ldw + k] = (float)t;
  }
}

// Y[M, Fp] (half) = [X[M, F] (fp32) | zeros]
__global__ void __launch_bounds__(256) k_pad_cast(const float* __restrict__ X, int64_t ldx, int F,
                                                  __half* __restrict__ Y, int Fp, int64_t M) {
  const int pieces = Fp / 8;
  const int64_t total = M * pieces;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t row = i / pieces;
    const int c0 = 8 * (int)(i % pieces);
    h8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (c0 + e < F) ? (_Float16)X[row * ldx + c0 + e] : (_Float16)0.f;
    *reinterpret_cast<uint4*>(Y + row * Fp + c0) = __builtin_bit_cast(uint4, v);
  }
}

// B fragments + the waves' slabs; the statistics fold reuses
// the panel, which must then hold HG_NW x 2 x cols doubles
size_t hg_lds_bytes(int K, int nt, int stats) {
  size_t panel = (size_t)K * nt * 64;
  if (stats && panel < (size_t)HG_NW * 2 * 32 * nt * sizeof(double)) panel = (size_t)HG_NW * 2 * 32 * nt * sizeof(double);
  return panel + (size_t)HG_NW * 32 * HG_SLD * sizeof(__half);
}

template <int NT, bool BT, int STATS>
bool hg_attr() {
  static bool done[CGNN_MAX_DEVICES] = {};
  bool& d = done[cgnn_device_ordinal()];
  if (!d) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_hgemm<NT, BT, STATS>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return false;
    d = true;
  }
  return true;
}

template <bool BT, int STATS>
int hg_launch(const __half* X, int64_t ldx, int K, const float* W, int ldw, int Kw, const float* bias,
              __half* Y, int64_t ldy, int64_t M, int cols, hipStream_t st, HgStats hs = HgStats{}) {
  const int nt = cols / 32;
  const int grid = cgnn_fused_grid();
  const size_t lds = hg_lds_bytes(K, nt, STATS);
  if (lds > 160 * 1024) return CGNN_EUNSUPPORTED;
#define HG_CASE(NTV)                                                                              \
  case NTV:                                                                                       \
    if (!hg_attr<NTV, BT, STATS>()) return CGNN_ELAUNCH;                                          \
    k_hgemm<NTV, BT, STATS><<<grid, HG_THR, lds, st>>>(X, ldx, K, W, ldw, Kw, bias, Y, ldy, M, hs); \
    break;
  switch (nt) {
    HG_CASE(4) HG_CASE(8)
    case 2:
      if (STATS) return CGNN_EUNSUPPORTED;
      if (!hg_attr<2, BT, 0>()) return CGNN_ELAUNCH;
      k_hgemm<2, BT, 0><<<grid, HG_THR, lds, st>>>(X, ldx, K, W, ldw, Kw, bias, Y, ldy, M, hs);
      break;
    default: return CGNN_EUNSUPPORTED;
  }
#undef HG_CASE
  CGNN_CHECK_LAUNCH();
  return CGNN_OK;
}

bool hg_shape_ok(int K, int cols) {
  return (cols == 64 || cols == 128 || cols == 256) && K >= 32 && K % 32 == 0 &&
         (int64_t)K * (cols / 32) * 64 <= HG_MAX_PANEL;
}

// tile shape (n width, k width) of the weight-gradient kernel for [N x K], rows per run and runs
void wg_plan(int64_t M, int N, int K, int* tnw, int* tkw, int* nruns, int64_t* per) {
  *tnw = N % 128 == 0 ? 128 : 64;
  *tkw = K % 128 == 0 ? 128 : 64;
  const int ntiles = (N / *tnw) * (K / *tkw);
  int r = cgnn_fused_grid() / ntiles;
  if (r < 1) r = 1;
  const int64_t need = (M + WG_ROWS - 1) / WG_ROWS;               // never more runs than stages of rows
  if (r > need) r = (int)(need > 0 ? need : 1);
  *nruns = r;
  *per = ((M + r - 1) / r + WG_ROWS - 1) / WG_ROWS * WG_ROWS;
}

}  // namespace

extern "C" {

int cgnn_linear_fwd_f16(const void* X, int64_t ldx, int32_t K, const float* W, int32_t ldw, int32_t Kw,
                        const float* bias, void* Y, int64_t ldy, int64_t M, int32_t N, void* stream) {
  if (M < 0 || K <= 0 || N <= 0 || Kw <= 0 || Kw > K || ldw < Kw || ldx < K || ldy < N) return CGNN_EINVAL;
  if (!hg_shape_ok(K, N) || ldx % 8 || ldy % 8 || (reinterpret_cast<uintptr_t>(X) & 15) ||
      (reinterpret_cast<uintptr_t>(Y) & 15) || (reinterpret_cast<uintptr_t>(W) & 15))
    return CGNN_EUNSUPPORTED;
  if (M == 0) return CGNN_OK;
  if (!X || !W || !Y) return CGNN_EINVAL;
  return hg_launch<false, 0>(static_cast<const __half*>(X), ldx, K, W, ldw, Kw, bias, static_cast<__half*>(Y), ldy, M,
                             N, cgnn_stream(stream));
}

int cgnn_linear_fwd_stats_f16(const void* X, int64_t ldx, int32_t K, const float* W, int32_t ldw, int32_t Kw,
                              const float* bias, void* Y, int64_t ldy, int64_t M, int32_t N, double* stat_slab, int64_t stat_slab_bytes,
                              void* stream) {
  if (M < 0 || K <= 0 || N <= 0 || Kw <= 0 || Kw > K || ldw < Kw || ldx < K || ldy < N || !stat_slab) return CGNN_EINVAL;
  if (!hg_shape_ok(K, N) || N < 128 || ldx % 8 || ldy % 8 || (reinterpret_cast<uintptr_t>(X) & 15) ||
      (reinterpret_cast<uintptr_t>(Y) & 15) || (reinterpret_cast<uintptr_t>(W) & 15))
    return CGNN_EUNSUPPORTED;
  if (!X || !W || !Y) return CGNN_EINVAL;
  CGNN_NEED_BYTES(stat_slab, stat_slab_bytes, (int64_t)cgnn_fused_grid() * 2 * N * (int64_t)sizeof(double));
  HgStats hs{};
  hs.slab = stat_slab;
  return hg_launch<false, 1>(static_cast<const __half*>(X), ldx, K, W, ldw, Kw, bias, static_cast<__half*>(Y), ldy, M,
                             N, cgnn_stream(stream), hs);
}

int cgnn_linear_bwd_input_f16(const void* dY, int64_t lddy, const float* W, int32_t ldw, void* dX,
                              int64_t lddx, int64_t M, int32_t N, int32_t K, void* stream) {
  if (M < 0 || K <= 0 || N <= 0 || ldw < K || lddy < N || lddx < K) return CGNN_EINVAL;
  if (!hg_shape_ok(N, K) || ldw % 2 || lddy % 8 || lddx % 8 || (reinterpret_cast<uintptr_t>(dY) & 15) ||
      (reinterpret_cast<uintptr_t>(dX) & 15) || (reinterpret_cast<uintptr_t>(W) & 7))
    return CGNN_EUNSUPPORTED;
  if (M == 0) return CGNN_OK;
  if (!dY || !W || !dX) return CGNN_EINVAL;
  return hg_launch<true, 0>(static_cast<const __half*>(dY), lddy, N, W, ldw, N, nullptr, static_cast<__half*>(dX),
                            lddx, M, K, cgnn_stream(stream));
}

int64_t cgnn_linear_bwd_weight_f16_workspace_bytes(int64_t M, int32_t N, int32_t K) {
  if (M < 0 || N <= 0 || K <= 0 || N % 64 || K % 64 || N > 256 || K > 256) return CGNN_EUNSUPPORTED;
  int nruns, tnw, tkw;
  int64_t per;
  wg_plan(M, N, K, &tnw, &tkw, &nruns, &per);
  return cgnn_align_up((int64_t)nruns * N * K * (int64_t)sizeof(float), 256);
}

int cgnn_linear_bwd_weight_f16(const void* dY, int64_t lddy, const void* X, int64_t ldx, float* dW,
                               int32_t ldw, int32_t Kw, int64_t M, int32_t N, int32_t K, void* slab, int64_t slab_bytes,
                               void* stream) {
  if (M < 0 || N <= 0 || K <= 0 || Kw <= 0 || Kw > K || ldw < Kw || lddy < N || ldx < K) return CGNN_EINVAL;
  if (N % 64 || K % 64 || N > 256 || K > 256 || lddy % 8 || ldx % 8 || (reinterpret_cast<uintptr_t>(dY) & 15) ||
      (reinterpret_cast<uintptr_t>(X) & 15))
    return CGNN_EUNSUPPORTED;
  if (!dW || !slab) return CGNN_EINVAL;
  if (M > 0 && (!dY || !X)) return CGNN_EINVAL;
  hipStream_t st = cgnn_stream(stream);
  int nruns, tnw, tkw;
  int64_t per;
  wg_plan(M, N, K, &tnw, &tkw, &nruns, &per);
  CGNN_NEED_BYTES(slab, slab_bytes, (int64_t)nruns * N * K * (int64_t)sizeof(float));
  const int grid = nruns * (N / tnw) * (K / tkw);                  // every (run, tile)
  const __half* a = static_cast<const __half*>(dY);
  const __half* b = static_cast<const __half*>(X);
  float* sl = static_cast<float*>(slab);
  if (tnw == 128 && tkw == 128)
    k_hgemm_wgrad<2, 4, 2><<<grid, HG_THR, 0, st>>>(a, lddy, b, ldx, sl, M, N, K, nruns, per);
  else if (tnw == 128)
    k_hgemm_wgrad<1, 4, 2><<<grid, HG_THR, 0, st>>>(a, lddy, b, ldx, sl, M, N, K, nruns, per);
  else if (tkw == 128)
    k_hgemm_wgrad<1, 2, 4><<<grid, HG_THR, 0, st>>>(a, lddy, b, ldx, sl, M, N, K, nruns, per);
  else
    k_hgemm_wgrad<1, 2, 2><<<grid, HG_THR, 0, st>>>(a, lddy, b, ldx, sl, M, N, K, nruns, per);
  CGNN_CHECK_LAUNCH();
  const int64_t elems = (int64_t)N * K;
  k_hgemm_fold<<<(unsigned)((elems + 15) / 16), 256, 0, st>>>(static_cast<const float*>(slab), nruns, N, K, Kw, dW,
                                                              ldw);
  CGNN_CHECK_LAUNCH();
  return CGNN_OK;
}

int cgnn_pad_cast_f16(const float* X, int64_t ldx, int32_t F, void* Y, int32_t Fp, int64_t M, void* stream) {
  if (M < 0 || F <= 0 || Fp < F || Fp % 8 || ldx < F) return CGNN_EINVAL;
  if (reinterpret_cast<uintptr_t>(Y) & 15) return CGNN_EUNSUPPORTED;
  if (M == 0) return CGNN_OK;
  if (!X || !Y) return CGNN_EINVAL;
  const int64_t total = M * (Fp / 8);
  const int64_t want = (total + 255) / 256;
  const unsigned grid = (unsigned)(want < 4096 ? want : 4096);
  k_pad_cast<<<grid, 256, 0, cgnn_stream(stream)>>>(X, ldx, F, static_cast<__half*>(Y), Fp, M);
  CGNN_CHECK_LAUNCH();
  return CGNN_OK;
}

}  // extern "C"
