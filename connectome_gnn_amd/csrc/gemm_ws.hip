// gemm_ws.hip -- weight-stationary projection kernels for the wide hidden layers (gfx950, fp32).
//
// Same arithmetic as gemm.hip (models.py:111, :151-152 and autograd's backward of them), other
// execution shape.  The layer shapes are tall-skinny: M = every node of the batch (10^5..10^6),
// N, K <= 256.  gemm.hip stages both operands through LDS per 32-wide K step with two barriers
// per step and reaches ~35 % of the fp32 matrix-core peak.  Here
//
//   fwd / bwd_input : the whole weight panel (<= 128 KB) is parked in LDS once per persistent
//                     workgroup; every wave then streams ITS OWN 32 rows of the activation
//                     straight from HBM into registers as the MFMA A operand (lane = row, 16-byte
//                     loads along K, two chunks in flight) -- no barrier in the main loop, the B
//                     operand of four 32x32 output tiles is one conflict-free ds_read_b128.
//   bwd_weight      : reduction over M; both operands stream from HBM/L1 directly into MFMA
//                     registers (lane = column), nothing goes through LDS, each wave owns a
//                     [N x 32] strip of dW; per-workgroup partials are reduced in fixed order.
//
// v_mfma_f32_32x32x2_f32 operand maps (cdna_hip_programming.md section 3):
//   A: lane l holds A[i = l&31][k = l>>5]      B: lane l holds B[k = l>>5][j = l&31]
//   C/D reg r of lane l: row = (r&3) + 8*(r>>2) + 4*(l>>5), col = l&31
// The reduction index and the tile->column map are permuted freely: the two lane halves of a row
// read alternating 16-byte pieces of that row (k = 8*i + 4*h + e), and output
// tile t of a group holds the columns {4*j + t}, so a lane ends up with 4 consecutive columns
// of each of its rows = 16-byte stores.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int WS_NW = 8;                 // waves per workgroup of bwd_weight (one workgroup per CU)
constexpr int WS_THR = WS_NW * 64;
#ifndef WS_FWD_NW
#define WS_FWD_NW 8
#endif
constexpr int FW_NW = WS_FWD_NW;         // waves per workgroup of fwd / bwd_input
constexpr int FW_THR = FW_NW * 64;
constexpr int WS_LDS_FLOATS = 32768;     // 128 KB weight panel
#ifndef WS_UNR
#define WS_UNR 4
#endif
constexpr int UNR = WS_UNR;              // float4 loads of the A row per chunk (4*UNR k-steps)

__device__ __forceinline__ float4 ldg4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// ------------------------------------------------------------------------------------ forward
// Y[M, N] = act([X1 | X2] W^T + b),  W [N, K1+K2] row-major, N = 32*NT.
template <int NT>
__global__ void __launch_bounds__(FW_THR) k_ws_fwd(
    const float* __restrict__ X1, int64_t ldx1, int K1, const float* __restrict__ X2,
    int64_t ldx2, int K2, const float* __restrict__ W, const float* __restrict__ bias, int relu,
    float* __restrict__ Y, int64_t ldy, int64_t M, double* __restrict__ stat_slab) {
  constexpr int N = 32 * NT;
  __shared__ __attribute__((aligned(16))) float Wl[WS_LDS_FLOATS];   // Wl[k][n] = W[n][k]
  const int K = K1 + K2;
  // (loads batched 8 deep: one exposed L2 round trip per batch, not per element)
  for (int idx0 = threadIdx.x; idx0 < N * (K / 4); idx0 += 8 * FW_THR) {
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = idx0 + u * FW_THR;
      if (idx < N * (K / 4)) v[u] = ldg4(W + (int64_t)(idx % N) * K + 4 * (idx / N));
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = idx0 + u * FW_THR;
      if (idx < N * (K / 4)) {
        const int n = idx % N, k4 = idx / N;
        Wl[(4 * k4 + 0) * N + n] = v[u].x; Wl[(4 * k4 + 1) * N + n] = v[u].y;
        Wl[(4 * k4 + 2) * N + n] = v[u].z; Wl[(4 * k4 + 3) * N + n] = v[u].w;
      }
    }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 31, kh = lane >> 5;
  // reduction index of lane half kh at load u of chunk c: 8*(c*UNR + u) + 4*kh + e -- the two
  // lanes of a row read ADJACENT 16-byte pieces, so a wave load touches 32 lines, not 64 (the
  // L1 request rate, ~0.3 lines/clk/CU for such partial-line accesses, is what the A stream costs)
  const int c0 = 4 * kh;
  const int nchunks = (K / 8) / UNR;
  const int64_t nrb = (M + 31) / 32;
  const int64_t stride = (int64_t)gridDim.x * FW_NW;
  float bv[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) bv[t] = bias ? bias[NT * j + t] : 0.f;

  auto row_ptr = [&](int64_t rb, int kc) -> const float* {
    int64_t row = rb * 32 + j;
    if (row >= M) row = M - 1;                       // clamp: loads stay in bounds, rows unused
    return kc < K1 ? X1 + row * ldx1 + kc : X2 + row * ldx2 + (kc - K1);
  };

  int64_t rb = (int64_t)blockIdx.x * FW_NW + wave;
  float4 cur[UNR], nxt[UNR];
  double s1[NT], s2[NT];                            // column sums of Y, Y^2 (BatchNorm statistics)
#pragma unroll
  for (int t = 0; t < NT; ++t) s1[t] = s2[t] = 0.0;
  if (rb < nrb) {
#pragma unroll
    for (int u = 0; u < UNR; ++u) cur[u] = ldg4(row_ptr(rb, c0 + 8 * u));
  }
  for (; rb < nrb; rb += stride) {
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = f32x16{0};
    for (int c = 0; c < nchunks; ++c) {
      // request the next chunk (of this row block, or the first one of the wave's next block)
      const bool last = c + 1 == nchunks;
      const int64_t rbn = last ? rb + stride : rb;
      const int cn = last ? 0 : c + 1;
      if (rbn < nrb) {
#pragma unroll
        for (int u = 0; u < UNR; ++u) nxt[u] = ldg4(row_ptr(rbn, c0 + 8 * (cn * UNR + u)));
      }
      const float* wl = Wl + (int64_t)(c0 + 8 * c * UNR) * N + NT * j;
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const float av[4] = {cur[u].x, cur[u].y, cur[u].z, cur[u].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float* wp = wl + (8 * u + e) * N;
          if (NT == 4) {
            const float4 b = *reinterpret_cast<const float4*>(wp);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], b.x, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], b.y, acc[1], 0, 0, 0);
            acc[NT > 2 ? 2 : 0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], b.z, acc[NT > 2 ? 2 : 0], 0, 0, 0);
            acc[NT > 3 ? 3 : 0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], b.w, acc[NT > 3 ? 3 : 0], 0, 0, 0);
          } else {
            const float2 b = *reinterpret_cast<const float2*>(wp);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], b.x, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], b.y, acc[1], 0, 0, 0);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u) cur[u] = nxt[u];
    }
    // epilogue: bias, activation, 16-byte (NT = 4) / 8-byte (NT = 2) stores
    float p1[NT], p2[NT];                           // this block's 16 rows: fp32 partials
#pragma unroll
    for (int t = 0; t < NT; ++t) p1[t] = p2[t] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int64_t row = rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
      if (row < M) {
        float v[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          v[t] = acc[t][r] + bv[t];
          if (relu) v[t] = fmaxf(v[t], 0.f);
          p1[t] += v[t];
          p2[t] = fmaf(v[t], v[t], p2[t]);
        }
        float* yp = Y + row * ldy + NT * j;
        if (NT == 4) *reinterpret_cast<float4*>(yp) = make_float4(v[0], v[1], v[NT > 2 ? 2 : 0], v[NT > 3 ? 3 : 0]);
        else *reinterpret_cast<float2*>(yp) = make_float2(v[0], v[1]);
      }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) { s1[t] += (double)p1[t]; s2[t] += (double)p2[t]; }
  }
  if (stat_slab) {
    // lane (j, kh) holds columns NT*j + t; fold the two row halves, then the waves (fixed order)
    double* red = reinterpret_cast<double*>(Wl);   // [FW_NW][2N], the weight panel is dead now
    __syncthreads();
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      s1[t] += __shfl_xor(s1[t], 32, 64);
      s2[t] += __shfl_xor(s2[t], 32, 64);
      if (kh == 0) {
        red[wave * 2 * N + NT * j + t] = s1[t];
        red[wave * 2 * N + N + NT * j + t] = s2[t];
      }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 2 * N; e += FW_THR) {
      double tsum = 0.0;
#pragma unroll
      for (int w2 = 0; w2 < FW_NW; ++w2) tsum += red[w2 * 2 * N + e];
      stat_slab[(int64_t)blockIdx.x * 2 * N + e] = tsum;
    }
  }
}

// ----------------------------------------------------------------------------- backward input
// dX[M, K] = dY[M, N] W[N, k0:k0+K];  K = 32*NT output columns, reduction over N.
// Output tile t <-> columns 128*(t/4) + 4*j + t%4 (NT >= 4) or 2*j + t (NT == 2).
template <int NT>
__global__ void __launch_bounds__(FW_THR) k_ws_bwd_input(
    const float* __restrict__ dY, int64_t lddy, const float* __restrict__ W, int ldw, int k0,
    float* __restrict__ dX, int64_t lddx, int64_t M, int N) {
  constexpr int K = 32 * NT;
  constexpr int VW = NT >= 4 ? 4 : 2;                // columns per lane per group
  constexpr int NG = NT / VW;                        // column groups of 32*VW
  __shared__ __attribute__((aligned(16))) float Wl[WS_LDS_FLOATS];   // Wl[n][kc] = W[n][k0+kc]
  for (int idx0 = threadIdx.x; idx0 < N * (K / 4); idx0 += 8 * FW_THR) {
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = idx0 + u * FW_THR;
      if (idx < N * (K / 4)) v[u] = ldg4(W + (int64_t)(idx / (K / 4)) * ldw + k0 + 4 * (idx % (K / 4)));
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = idx0 + u * FW_THR;
      if (idx < N * (K / 4)) *reinterpret_cast<float4*>(Wl + 4 * idx) = v[u];
    }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 31, nh = lane >> 5;
  const int c0 = 4 * nh;                             // same interleaved reduction map as k_ws_fwd
  const int nchunks = (N / 8) / UNR;
  const int64_t nrb = (M + 31) / 32;
  const int64_t stride = (int64_t)gridDim.x * FW_NW;

  auto row_ptr = [&](int64_t rb) -> const float* {
    int64_t row = rb * 32 + j;
    if (row >= M) row = M - 1;
    return dY + row * lddy + c0;
  };

  int64_t rb = (int64_t)blockIdx.x * FW_NW + wave;
  float4 cur[UNR], nxt[UNR];
  if (rb < nrb) {
    const float* p = row_ptr(rb);
#pragma unroll
    for (int u = 0; u < UNR; ++u) cur[u] = ldg4(p + 8 * u);
  }
  for (; rb < nrb; rb += stride) {
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = f32x16{0};
    for (int c = 0; c < nchunks; ++c) {
      const bool last = c + 1 == nchunks;
      const int64_t rbn = last ? rb + stride : rb;
      const int cn = last ? 0 : c + 1;
      if (rbn < nrb) {
        const float* p = row_ptr(rbn) + 8 * cn * UNR;
#pragma unroll
        for (int u = 0; u < UNR; ++u) nxt[u] = ldg4(p + 8 * u);
      }
      const float* wl = Wl + (c0 + 8 * c * UNR) * K + VW * j;
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const float av[4] = {cur[u].x, cur[u].y, cur[u].z, cur[u].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float* wp = wl + (8 * u + e) * K;
#pragma unroll
          for (int g = 0; g < NG; ++g) {
            if (VW == 4) {
              const float4 b = *reinterpret_cast<const float4*>(wp + 128 * g);
              acc[4 * g + 0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], b.x, acc[4 * g + 0], 0, 0, 0);
              acc[4 * g + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], b.y, acc[4 * g + 1], 0, 0, 0);
              acc[(4 * g + 2) % NT] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], b.z, acc[(4 * g + 2) % NT], 0, 0, 0);
              acc[(4 * g + 3) % NT] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], b.w, acc[(4 * g + 3) % NT], 0, 0, 0);
            } else {
              const float2 b = *reinterpret_cast<const float2*>(wp);
              acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], b.x, acc[0], 0, 0, 0);
              acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], b.y, acc[1], 0, 0, 0);
            }
          }
        }
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u) cur[u] = nxt[u];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int64_t row = rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * nh;
      if (row < M) {
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          float* xp = dX + row * lddx + 128 * g + VW * j;
          if (VW == 4)
            *reinterpret_cast<float4*>(xp) = make_float4(acc[4 * g][r], acc[4 * g + 1][r], acc[(4 * g + 2) % NT][r],
                                                         acc[(4 * g + 3) % NT][r]);
          else
            *reinterpret_cast<float2*>(xp) = make_float2(acc[0][r], acc[1][r]);
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------- backward weight
// partial[p][n][kc] = sum_{m in chunk p} dY[m][n] X[m][kc];  N = 32*NTN, K = 32*TK.
// Wave w owns the [N x 32] strip kt = w % TK of dW and the sub-chunk w / TK of the workgroup's
// rows.  A operand = dY^T: lane (i, mh) loads dY[m][NTN*i .. +NTN) (tile t <-> n = NTN*i + t);
// B operand = X: lane (j, mh) loads X[m][32*kt + j].  Row m = 2*s + mh at step s.
constexpr int WU = 8;                                // steps (row pairs) per software stage

template <int NTN>
__global__ void __launch_bounds__(WS_THR) k_ws_bwd_weight(
    const float* __restrict__ dY, int64_t lddy, const float* __restrict__ X1, int64_t ldx1, int K1,
    const float* __restrict__ X2, int64_t ldx2, float* __restrict__ slab, int64_t M, int K,
    int TK) {
  constexpr int N = 32 * NTN;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 31, mh = lane >> 5;
  const int kt = cgnn_uniform(wave % TK), msub = cgnn_uniform(wave / TK), MS = WS_NW / TK;
  // rows of this wave: the batch is cut into gridDim.x * MS contiguous pieces of whole row pairs
  const int64_t pieces = (int64_t)gridDim.x * MS;
  const int64_t per = ((M + pieces - 1) / pieces + 1) & ~(int64_t)1;
  const int64_t piece = (int64_t)blockIdx.x * MS + msub;
  const int64_t mbeg = min(M, piece * per), mend = min(M, mbeg + per);
  const float* ap = dY + NTN * j;
  const bool second = 32 * kt >= K1;              // wave-uniform: which K-panel this strip is in
  const float* bp = second ? X2 + (32 * kt - K1) + j : X1 + 32 * kt + j;
  const int64_t ldx = second ? ldx2 : ldx1;
  f32x16 acc[NTN];
#pragma unroll
  for (int t = 0; t < NTN; ++t) acc[t] = f32x16{0};

  float av[WU][NTN], bvv[WU], an[WU][NTN], bn[WU];
  auto load = [&](int64_t m0, float (&a)[WU][NTN], float (&b)[WU]) {
#pragma unroll
    for (int s = 0; s < WU; ++s) {
      const int64_t m = m0 + 2 * s + mh;
      if (m < mend) {
        if (NTN == 4) {
          const float4 v = ldg4(ap + m * lddy);
          a[s][0] = v.x; a[s][1] = v.y; a[s][NTN > 2 ? 2 : 0] = v.z; a[s][NTN > 3 ? 3 : 0] = v.w;
        } else {
          const float2 v = *reinterpret_cast<const float2*>(ap + m * lddy);
          a[s][0] = v.x; a[s][1] = v.y;
        }
        b[s] = bp[m * ldx];
      } else {
#pragma unroll
        for (int t = 0; t < NTN; ++t) a[s][t] = 0.f;
        b[s] = 0.f;
      }
    }
  };
  if (mbeg < mend) load(mbeg, av, bvv);
  for (int64_t m0 = mbeg; m0 < mend; m0 += 2 * WU) {
    if (m0 + 2 * WU < mend) load(m0 + 2 * WU, an, bn);
#pragma unroll
    for (int s = 0; s < WU; ++s)
#pragma unroll
      for (int t = 0; t < NTN; ++t)
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s][t], bvv[s], acc[t], 0, 0, 0);
#pragma unroll
    for (int s = 0; s < WU; ++s) {
#pragma unroll
      for (int t = 0; t < NTN; ++t) av[s][t] = an[s][t];
      bvv[s] = bn[s];
    }
  }
  float* out = slab + piece * (int64_t)N * K + 32 * kt + j;
#pragma unroll
  for (int t = 0; t < NTN; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = (r & 3) + 8 * (r >> 2) + 4 * mh;
      out[(int64_t)(NTN * i + t) * K] = acc[t][r];
    }
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

constexpr int64_t WS_MIN_ROWS = 4096;    // below this the launch-per-tile kernels of gemm.hip win

}  // namespace

// Internal hooks for gemm.hip: return true if the weight-stationary kernel was launched.
bool cgnn_ws_linear_fwd(const float* X1, int64_t ldx1, int K1, const float* X2, int64_t ldx2,
                        int K2, const float* W, const float* bias, int relu, float* Y,
                        int64_t ldy, int64_t M, int N, double* stat_slab, hipStream_t st) {
  const int K = K1 + K2;
  if (M < WS_MIN_ROWS || (N != 64 && N != 128) || K % 32 || K1 % 4 || K2 % 4 ||
      (int64_t)N * K > WS_LDS_FLOATS)
    return false;
  if (ldx1 % 4 || (K2 && ldx2 % 4) || ldy % 4 || !aligned16(X1) || (K2 && !aligned16(X2)) ||
      !aligned16(W) || !aligned16(Y))
    return false;
  const int grid = cgnn_fused_grid();
  if (N == 128)
    k_ws_fwd<4><<<grid, FW_THR, 0, st>>>(X1, ldx1, K1, X2, ldx2, K2, W, bias, relu, Y, ldy, M, stat_slab);
  else
    k_ws_fwd<2><<<grid, FW_THR, 0, st>>>(X1, ldx1, K1, X2, ldx2, K2, W, bias, relu, Y, ldy, M, stat_slab);
  return true;
}

bool cgnn_ws_linear_bwd_input(const float* dY, int64_t lddy, const float* W, int ldw, int k0,
                              float* dX, int64_t lddx, int64_t M, int N, int K, hipStream_t st) {
  if (M < WS_MIN_ROWS || (K != 64 && K != 128 && K != 256) || N % 32 || N > 256 ||
      (int64_t)N * K > WS_LDS_FLOATS)
    return false;
  if (lddy % 4 || lddx % 4 || ldw % 4 || k0 % 4 || !aligned16(dY) || !aligned16(W) || !aligned16(dX))
    return false;
  const int grid = cgnn_fused_grid();
  switch (K) {
    case 64: k_ws_bwd_input<2><<<grid, FW_THR, 0, st>>>(dY, lddy, W, ldw, k0, dX, lddx, M, N); break;
    case 128: k_ws_bwd_input<4><<<grid, FW_THR, 0, st>>>(dY, lddy, W, ldw, k0, dX, lddx, M, N); break;
    default: k_ws_bwd_input<8><<<grid, FW_THR, 0, st>>>(dY, lddy, W, ldw, k0, dX, lddx, M, N); break;
  }
  return true;
}

// number of fp32 [N x K] partials the weight-stationary bwd_weight writes (0 = not eligible)
int64_t cgnn_ws_bwd_weight_partials(int64_t M, int N, int K) {
  if (M < WS_MIN_ROWS || (N != 64 && N != 128)) return 0;
  if (K != 32 && K != 64 && K != 128 && K != 256) return 0;
  return (int64_t)cgnn_fused_grid() * (WS_NW / (K / 32));
}

bool cgnn_ws_linear_bwd_weight(const float* dY, int64_t lddy, const float* X1, int64_t ldx1, int K1,
                               const float* X2, int64_t ldx2, int K2, float* slab, int64_t M,
                               int N, hipStream_t st) {
  const int K = K1 + K2;
  if (cgnn_ws_bwd_weight_partials(M, N, K) == 0) return false;
  if (lddy % 4 || !aligned16(dY) || (K2 > 0 && K1 % 32)) return false;
  const int grid = cgnn_fused_grid();
  if (N == 128)
    k_ws_bwd_weight<4><<<grid, WS_THR, 0, st>>>(dY, lddy, X1, ldx1, K1, X2, ldx2, slab, M, K, K / 32);
  else
    k_ws_bwd_weight<2><<<grid, WS_THR, 0, st>>>(dY, lddy, X1, ldx1, K1, X2, ldx2, slab, M, K, K / 32);
  return true;
}
