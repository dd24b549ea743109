// gemm_ws.hip -- weight-stationary projection kernels for the wide hidden layers (gfx950).
//
// Same arithmetic as gemm.hip (models.py:111, :151-152 and autograd's backward of them), other
// execution shape.  The layer shapes are tall-skinny: M = every node of the batch (10^5..10^6),
// N, K <= 256.  Products are exact fp32 products computed on the bf16 matrix pipe (split_bf16.h:
// three truncation pieces per operand, six partial products, fp32 accumulation), 2.7x the rate of
// v_mfma_f32_32x32x2_f32, which leaves these kernels bound by the activation stream from HBM.
//
//   k_ws (fwd, bwd_input) : the weight panel, already split into bf16 pieces and laid out as MFMA
//                     B fragments, is parked in LDS once per persistent workgroup (<= 96 KB); every
//                     wave then streams ITS OWN 32 rows of the activation straight from HBM into
//                     registers (lane = row, 16-byte loads along K, one chunk ahead), splits them
//                     and feeds the MFMA A operand -- no barrier in the main loop, a B fragment is
//                     one conflict-free ds_read_b128.  When the split panel of all output columns
//                     exceeds the LDS, the columns are cut into G groups; workgroups b and b + 8
//                     (same XCD, same L2) take the same rows for different groups at the same
//                     time, so the second read of a row block is an L2 hit.
//   k_ws_bwd_weight : reduction over M; both operands stream from HBM/L1 directly into registers
//                     (lane = column, 8 consecutive rows each), nothing goes through LDS, each
//                     wave owns a [N x 32] strip of dW; one partial per workgroup, reduced in
//                     fixed order.
//
// v_mfma_f32_32x32x16_bf16 operand maps (cdna_hip_programming.md section 3):
//   A: lane l holds A[i = l&31][k = 8*(l>>5) + e]     B: lane l holds B[k = 8*(l>>5) + e][j = l&31]
//   C/D reg r of lane l: row = (r&3) + 8*(r>>2) + 4*(l>>5), col = l&31
// The reduction index and the tile->column map are permuted freely: the two lane halves of a row
// read alternating 16-byte pieces of that row (k = 16*s + 8*(e>>2) + 4*h + (e&3) at step s), and
// output tile t of a group holds the columns {NT*j + t}, so a lane ends up with NT consecutive
// columns of each of its rows = 16-byte (8-byte) stores.
#include "common.h"
#include "split_bf16.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int WS_NW = 8;                 // waves per workgroup of bwd_weight (one workgroup per CU)
constexpr int WS_THR = WS_NW * 64;
#define WS_FWD_NW 12
constexpr int FW_NW = WS_FWD_NW;         // waves per workgroup of fwd / bwd_input
constexpr int FW_THR = FW_NW * 64;
constexpr int WS_PANEL_FRAGS = 6144;     // 96 KB of 16-byte B fragments: (K/16) * NT * 3 * 64 <= this
constexpr int WS_SLD = 36;               // row stride (floats) of a wave's 32 x 32 staging slab

__device__ __forceinline__ float4 ldg4(const float* p) { return *reinterpret_cast<const float4*>(p); }
#define WS_LD ldg4
#define WS_LDW ldg4
#define WS_LDS(p) (*(p))

__device__ __forceinline__ f32x16 mfma32(const uint4& a, const uint4& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b),
                                                 c, 0, 0, 0);
}
// acc += A * B for split operands, smallest terms first
__device__ __forceinline__ f32x16 mfma32_split(const Split8& a, const uint4& bh, const uint4& bm,
                                               const uint4& bl, f32x16 acc) {
  acc = mfma32(a.l, bh, acc);
  acc = mfma32(a.h, bl, acc);
  acc = mfma32(a.m, bm, acc);
  acc = mfma32(a.m, bh, acc);
  acc = mfma32(a.h, bm, acc);
  acc = mfma32(a.h, bh, acc);
  return acc;
}

// ---------------------------------------------------------------------- forward / backward input
// Y[M, cols] = act([X1 | X2] B + b) for the column group of this workgroup, 32*NT columns.
//   BT = false (forward):        B[k][c] = W[c * ldw + k]      (Y = X W^T, W [N, K] row-major)
//   BT = true  (backward input): B[k][c] = W[k * ldw + c]      (dX = dY W)
// gridDim.x = G * (row groups); `pair` != 0: workgroups b and b + 8 share rows (see above).
template <int NT, bool BT>
__global__ void __launch_bounds__(FW_THR) k_ws(
    const float* __restrict__ X1, int64_t ldx1, int K1, const float* __restrict__ X2,
    int64_t ldx2, int K2, const float* __restrict__ W, int64_t ldw, const float* __restrict__ bias,
    int relu, float* __restrict__ Y, int64_t ldy, int64_t M, int N, int G, int pair,
    double* __restrict__ stat_slab) {
  constexpr int NL = 32 * NT;                         // columns of this workgroup
  __shared__ __attribute__((aligned(16))) uint4 Wl[WS_PANEL_FRAGS];   // [(s*NT + t)*3 + piece][lane]
  __shared__ __attribute__((aligned(16))) float stg_all[FW_NW * 32 * WS_SLD];
  const int K = K1 + K2;
  const int g = G == 1 ? 0 : (pair ? (blockIdx.x >> 3) % G : blockIdx.x % G);
  const int rgroup = G == 1 ? blockIdx.x : (pair ? (blockIdx.x & 7) + 8 * (blockIdx.x / (8 * G)) : blockIdx.x / G);
  const int nrgroups = gridDim.x / G;
  const int col0 = g * NL;
  const int nsteps = K / 16;
  if (BT) {
    for (int idx = threadIdx.x; idx < nsteps * 64; idx += FW_THR) {
      const int ln = idx & 63, s = idx >> 6, jj = ln & 31, hh = ln >> 5;
      float v[8][NT];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float* wp = W + (int64_t)(16 * s + 8 * (e >> 2) + 4 * hh + (e & 3)) * ldw + col0 + NT * jj;
        if (NT == 4) {
          const float4 q = ldg4(wp);
          v[e][0] = q.x; v[e][1] = q.y; v[e][NT > 2 ? 2 : 0] = q.z; v[e][NT > 3 ? 3 : 0] = q.w;
        } else {
          const float2 q = *reinterpret_cast<const float2*>(wp);
          v[e][0] = q.x; v[e][1] = q.y;
        }
      }
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const Split8 sp = split8(make_float4(v[0][t], v[1][t], v[2][t], v[3][t]),
                                 make_float4(v[4][t], v[5][t], v[6][t], v[7][t]));
        uint4* f = Wl + ((s * NT + t) * 3) * 64 + ln;
        f[0] = sp.h; f[64] = sp.m; f[128] = sp.l;
      }
    }
  } else {
    for (int idx = threadIdx.x; idx < nsteps * NT * 64; idx += FW_THR) {
      const int ln = idx & 63, st = idx >> 6, jj = ln & 31, hh = ln >> 5;
      const int t = st % NT, s = st / NT;
      const float* wp = W + (int64_t)(col0 + NT * jj + t) * ldw + 16 * s + 4 * hh;
      const Split8 sp = split8(ldg4(wp), ldg4(wp + 8));
      uint4* f = Wl + (st * 3) * 64 + ln;
      f[0] = sp.h; f[64] = sp.m; f[128] = sp.l;
    }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 31, kh = lane >> 5;
  // The A stream.  A chunk = 32 reduction indices = one 128-byte line per row.  Global loads are
  // line-shaped (load u: rows 8u + lane/8, 16-byte piece lane%8 -- 8 whole lines per wave load;
  // lane = row loads would touch 32 lines with 32 bytes each, and the L1 rate for such partial
  // lines, ~0.3 lines/clk/CU, then bounds the kernel); the chunk is turned into the MFMA layout
  // (lane = row j, pieces 8u + 4kh) through a wave-private LDS slab, no workgroup barrier.
  float* stg = stg_all + wave * (32 * WS_SLD);
  const int lrow = lane >> 3, lpc = lane & 7;
  const int nchunks = K / 32;
  const int64_t nrb = (M + 31) / 32;
  const int64_t stride = (int64_t)nrgroups * FW_NW;
  float bv[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) bv[t] = bias ? bias[col0 + NT * j + t] : 0.f;

  // two chunks ahead: the loads of a chunk are issued two compute phases before their use
  // (12 waves x 2 x 4 KB in flight per CU; one chunk ahead leaves the stream latency-bound)
  float4 cur[4], pf0[4], pf1[4];
  auto request = [&](float4 (&buf)[4], int64_t rb, int c) {
    const int kc = 32 * c;                            // chunk-uniform: which K-panel
    const float* base = kc < K1 ? X1 + kc : X2 + (kc - K1);
    const int64_t ld = kc < K1 ? ldx1 : ldx2;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      int64_t row = rb * 32 + 8 * u + lrow;
      if (row >= M) row = M - 1;                       // clamp: loads stay in bounds, rows unused
      buf[u] = WS_LD(base + row * ld + 4 * lpc);
    }
  };

  int64_t rb = (int64_t)rgroup * FW_NW + wave;       // block being computed, its chunk
  int c = 0;
  int64_t qrb = rb;                                  // next chunk to request
  int qc = 0;
  auto advance = [&](int64_t& b, int& ch) { if (++ch == nchunks) { ch = 0; b += stride; } };
  double s1[NT], s2[NT];                            // column sums of Y, Y^2 (BatchNorm statistics)
#pragma unroll
  for (int t = 0; t < NT; ++t) s1[t] = s2[t] = 0.0;
  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = f32x16{0};
  if (qrb < nrb) { request(pf0, qrb, qc); advance(qrb, qc); }
  if (qrb < nrb) { request(pf1, qrb, qc); advance(qrb, qc); }

  auto process = [&](float4 (&buf)[4]) {
#pragma unroll
    for (int u = 0; u < 4; ++u) *reinterpret_cast<float4*>(stg + (8 * u + lrow) * WS_SLD + 4 * lpc) = buf[u];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int u = 0; u < 4; ++u) cur[u] = *reinterpret_cast<const float4*>(stg + j * WS_SLD + 8 * u + 4 * kh);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (qrb < nrb) { request(buf, qrb, qc); advance(qrb, qc); }
    const uint4* wf = Wl + (c * 2 * NT * 3) * 64 + lane;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const Split8 a = split8(cur[2 * s], cur[2 * s + 1]);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const uint4* f = wf + ((s * NT + t) * 3) * 64;
        acc[t] = mfma32_split(a, f[0], f[64], f[128], acc[t]);
      }
    }
    if (c + 1 == nchunks) {
      // epilogue: bias, activation, 16-byte (NT = 4) / 8-byte (NT = 2) stores
      float p1[NT];                                   // this block's 16 rows: fp32 partial sums,
      double p2[NT];                                  // squares in fp64 (they cancel against mean^2)
#pragma unroll
      for (int t = 0; t < NT; ++t) { p1[t] = 0.f; p2[t] = 0.0; }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
        if (row < M) {
          float v[NT];
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            v[t] = acc[t][r] + bv[t];
            if (relu) v[t] = fmaxf(v[t], 0.f);
            if (!BT) { p1[t] += v[t]; p2[t] += (double)v[t] * v[t]; }
          }
          float* yp = Y + row * ldy + col0 + NT * j;
          if (NT == 4) *reinterpret_cast<float4*>(yp) = make_float4(v[0], v[1], v[NT > 2 ? 2 : 0], v[NT > 3 ? 3 : 0]);
          else *reinterpret_cast<float2*>(yp) = make_float2(v[0], v[1]);
        }
      }
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        if (!BT) { s1[t] += (double)p1[t]; s2[t] += p2[t]; }
        acc[t] = f32x16{0};
      }
    }
    advance(rb, c);
  };
  while (rb < nrb) {
    process(pf0);
    if (rb >= nrb) break;
    process(pf1);
  }
  if (!BT && stat_slab) {
    // lane (j, kh) holds columns col0 + NT*j + t; fold the two row halves, then the waves (fixed
    // order); columns of the other groups are zero in this workgroup's slab row
    double* red = reinterpret_cast<double*>(Wl);   // [FW_NW][2 NL], the weight panel is dead now
    __syncthreads();
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      s1[t] += __shfl_xor(s1[t], 32, 64);
      s2[t] += __shfl_xor(s2[t], 32, 64);
      if (kh == 0) {
        red[wave * 2 * NL + NT * j + t] = s1[t];
        red[wave * 2 * NL + NL + NT * j + t] = s2[t];
      }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 2 * N; e += FW_THR) {
      const int n = e % N - col0, which = e / N;
      double tsum = 0.0;
      if (n >= 0 && n < NL) {
#pragma unroll
        for (int w2 = 0; w2 < FW_NW; ++w2) tsum += red[w2 * 2 * NL + which * NL + n];
      }
      stat_slab[(int64_t)blockIdx.x * 2 * N + e] = tsum;
    }
  }
}

// ---------------------------------------------------------------------------- backward weight
// partial[p][n][kc] = sum_{m in chunk p} dY[m][n] X[m][kc];  N = 32*NTN, K = 32*TK.
// Wave w owns the [N x 32] strip kt = w % TK of dW and the sub-chunk w / TK of the workgroup's
// rows.  One step = 16 rows: lane (i, mh) holds rows m0 + 8*mh + e, e = 0..7, of
//   A operand = dY^T: dY[m][NTN*i .. +NTN)  (tile t <-> n = NTN*i + t)
//   B operand = X:    X[m][32*kt + i]
// both split in registers (five splits per step feed 6*NTN MFMAs).
template <int NTN>
__global__ void __launch_bounds__(WS_THR) k_ws_bwd_weight(
    const float* __restrict__ dY, int64_t lddy, const float* __restrict__ X1, int64_t ldx1, int K1,
    const float* __restrict__ X2, int64_t ldx2, float* __restrict__ slab, int64_t M, int K,
    int TK) {
  constexpr int N = 32 * NTN;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 31, mh = lane >> 5;
  const int kt = cgnn_uniform(wave % TK), msub = cgnn_uniform(wave / TK), MS = WS_NW / TK;
  // rows of this wave: the batch is cut into gridDim.x * MS contiguous pieces of whole steps
  const int64_t pieces = (int64_t)gridDim.x * MS;
  const int64_t per = ((M + pieces - 1) / pieces + 15) & ~(int64_t)15;
  const int64_t piece = (int64_t)blockIdx.x * MS + msub;
  const int64_t mbeg = min(M, piece * per), mend = min(M, mbeg + per);
  const float* ap = dY + NTN * j;
  const bool second = 32 * kt >= K1;              // wave-uniform: which K-panel this strip is in
  const float* bp = second ? X2 + (32 * kt - K1) + j : X1 + 32 * kt + j;
  const int64_t ldx = second ? ldx2 : ldx1;
  f32x16 acc[NTN];
#pragma unroll
  for (int t = 0; t < NTN; ++t) acc[t] = f32x16{0};

  float av[8][NTN], bvv[8], an[8][NTN], bn[8];
  auto load = [&](int64_t m0, float (&a)[8][NTN], float (&b)[8]) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int64_t m = m0 + 8 * mh + e;
      if (m < mend) {
        if (NTN == 4) {
          const float4 v = WS_LDW(ap + m * lddy);
          a[e][0] = v.x; a[e][1] = v.y; a[e][NTN > 2 ? 2 : 0] = v.z; a[e][NTN > 3 ? 3 : 0] = v.w;
        } else {
          const float2 v = *reinterpret_cast<const float2*>(ap + m * lddy);
          a[e][0] = v.x; a[e][1] = v.y;
        }
        b[e] = WS_LDS(bp + m * ldx);
      } else {
#pragma unroll
        for (int t = 0; t < NTN; ++t) a[e][t] = 0.f;
        b[e] = 0.f;
      }
    }
  };
  if (mbeg < mend) load(mbeg, av, bvv);
  for (int64_t m0 = mbeg; m0 < mend; m0 += 16) {
    if (m0 + 16 < mend) load(m0 + 16, an, bn);
    const Split8 b = split8(make_float4(bvv[0], bvv[1], bvv[2], bvv[3]),
                            make_float4(bvv[4], bvv[5], bvv[6], bvv[7]));
#pragma unroll
    for (int t = 0; t < NTN; ++t) {
      const Split8 a = split8(make_float4(av[0][t], av[1][t], av[2][t], av[3][t]),
                              make_float4(av[4][t], av[5][t], av[6][t], av[7][t]));
      acc[t] = mfma32_split(a, b.h, b.m, b.l, acc[t]);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
#pragma unroll
      for (int t = 0; t < NTN; ++t) av[e][t] = an[e][t];
      bvv[e] = bn[e];
    }
  }
  if (MS == 1) {
    float* out = slab + (int64_t)blockIdx.x * N * K + 32 * kt + j;
#pragma unroll
    for (int t = 0; t < NTN; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * mh;
        out[(int64_t)(NTN * i + t) * K] = acc[t][r];
      }
    return;
  }
  // several waves per strip (K < 256): fold their partials through LDS in fixed order, so that
  // the workgroup leaves ONE [N x K] partial (K = 32: 2048 partials of 16 KB took longer to reduce
  // than to produce)
  __shared__ float red[WS_NW * N * 32];
#pragma unroll
  for (int t = 0; t < NTN; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = (r & 3) + 8 * (r >> 2) + 4 * mh;
      red[wave * (N * 32) + (NTN * i + t) * 32 + j] = acc[t][r];
    }
  __syncthreads();
  float* out = slab + (int64_t)blockIdx.x * N * K;
  for (int e = threadIdx.x; e < N * K; e += WS_THR) {
    const int n = e / K, kc = e - n * K, strip = kc >> 5;
    float v = 0.f;
    for (int ms = 0; ms < MS; ++ms) v += red[(ms * TK + strip) * (N * 32) + n * 32 + (kc & 31)];
    out[e] = v;
  }
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

constexpr int64_t WS_MIN_ROWS = 4096;    // below this the launch-per-tile kernels of gemm.hip win

// column tiles per workgroup (NT) and column groups (G) for `cols` output columns and a reduction
// length `red`; false when no cut of the columns fits the LDS panel
bool ws_shape(int cols, int red, int* nt, int* g) {
  if (red % 16 || cols % 64) return false;
  int t = cols % 128 == 0 ? 4 : 2;
  if ((red / 16) * t * 3 * 64 > WS_PANEL_FRAGS) t = 2;
  if ((red / 16) * t * 3 * 64 > WS_PANEL_FRAGS) return false;
  *nt = t;
  *g = cols / (32 * t);
  return true;
}

template <bool BT>
bool ws_launch(const float* X1, int64_t ldx1, int K1, const float* X2, int64_t ldx2, int K2,
               const float* W, int64_t ldw, const float* bias, int relu, float* Y, int64_t ldy,
               int64_t M, int cols, double* stat_slab, hipStream_t st) {
  int nt = 0, G = 0;
  if (!ws_shape(cols, K1 + K2, &nt, &G)) return false;
  int grid = cgnn_fused_grid();
  if (G > grid) return false;
  const int pair = grid % (8 * G) == 0;
  if (!pair) grid -= grid % G;
  if (stat_slab && grid != cgnn_fused_grid()) return false;      // the slab has one row per workgroup
  if (nt == 4)
    k_ws<4, BT><<<grid, FW_THR, 0, st>>>(X1, ldx1, K1, X2, ldx2, K2, W, ldw, bias, relu, Y, ldy, M, cols, G,
                                          pair, stat_slab);
  else
    k_ws<2, BT><<<grid, FW_THR, 0, st>>>(X1, ldx1, K1, X2, ldx2, K2, W, ldw, bias, relu, Y, ldy, M, cols, G,
                                          pair, stat_slab);
  return true;
}

}  // namespace

// Internal hooks for gemm.hip: return true if the weight-stationary kernel was launched.
bool cgnn_ws_linear_fwd(const float* X1, int64_t ldx1, int K1, const float* X2, int64_t ldx2,
                        int K2, const float* W, const float* bias, int relu, float* Y,
                        int64_t ldy, int64_t M, int N, double* stat_slab, hipStream_t st) {
  const int K = K1 + K2;
  if (M < WS_MIN_ROWS || (N != 64 && N != 128 && N != 256) || K % 32 || K1 % 32 || K2 % 32) return false;
  if (ldx1 % 4 || (K2 && ldx2 % 4) || ldy % 4 || !aligned16(X1) || (K2 && !aligned16(X2)) ||
      !aligned16(W) || !aligned16(Y))
    return false;
  return ws_launch<false>(X1, ldx1, K1, X2, ldx2, K2, W, K, bias, relu, Y, ldy, M, N, stat_slab, st);
}

bool cgnn_ws_linear_bwd_input(const float* dY, int64_t lddy, const float* W, int ldw, int k0,
                              float* dX, int64_t lddx, int64_t M, int N, int K, hipStream_t st) {
  if (M < WS_MIN_ROWS || (K != 64 && K != 128 && K != 256) || N % 32 || N > 256) return false;
  if (lddy % 4 || lddx % 4 || ldw % 4 || k0 % 4 || !aligned16(dY) || !aligned16(W) || !aligned16(dX))
    return false;
  return ws_launch<true>(dY, lddy, N, nullptr, 0, 0, W + k0, ldw, nullptr, 0, dX, lddx, M, K, nullptr, st);
}

// number of fp32 [N x K] partials the weight-stationary bwd_weight writes (0 = not eligible)
int64_t cgnn_ws_bwd_weight_partials(int64_t M, int N, int K) {
  if (M < WS_MIN_ROWS || (N != 64 && N != 128)) return 0;
  if (K != 32 && K != 64 && K != 128 && K != 256) return 0;
  return (int64_t)cgnn_fused_grid();      // one per workgroup
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
