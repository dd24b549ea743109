// gemm.hip -- the dense feature projection on the CDNA4 matrix cores.
//
// fp32 in / fp32 accumulate with v_mfma_f32_32x32x2_f32 (bit-for-bit a k-ordered fmaf chain,
// so results stay within fp32 rounding of the reference's CPU sgemm; there is no TF32-like
// shortcut on gfx950 and none is wanted at a 1e-5 parity bar).
//
//   fwd        Y  = act(X1 W[:, :K1]^T + X2 W[:, K1:]^T + b)      models.py:111, :151-152
//   bwd_input  dX = dY W[:, k0:k0+K]                               autograd of the above
//   bwd_weight dW = dY^T X   (tall reduction over M = nodes)       autograd of the above
//
// Shapes are tall-skinny (M = nodes of the whole batch, up to millions; N, K <= 512), so every
// kernel tiles M across the grid and keeps the whole N/K extent of its tile on chip.
//
// MFMA 32x32x2 f32 operand maps (cdna_hip_programming.md section 3):
//   A: lane l holds A[i = l&31][k = l>>5]      B: lane l holds B[k = l>>5][j = l&31]
//   C/D reg r of lane l: row = (r&3) + 8*(r>>2) + 4*(l>>5), col = l&31
#include "common.h"

// weight-stationary kernels for the wide, tall shapes (gemm_ws.hip); false = not applicable
bool cgnn_ws_linear_fwd(const float* X1, int64_t ldx1, int K1, const float* X2, int64_t ldx2,
                        int K2, const float* W, const float* bias, int relu, float* Y,
                        int64_t ldy, int64_t M, int N, double* stat_slab, hipStream_t st);
bool cgnn_ws_linear_bwd_input(const float* dY, int64_t lddy, const float* W, int ldw, int k0,
                              float* dX, int64_t lddx, int64_t M, int N, int K, hipStream_t st);
int64_t cgnn_ws_bwd_weight_partials(int64_t M, int N, int K);
bool cgnn_ws_linear_bwd_weight(const float* dY, int64_t lddy, const float* X1, int64_t ldx1, int K1,
                               const float* X2, int64_t ldx2, int K2, float* slab, int64_t M,
                               int N, hipStream_t st);

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128;   // rows of the output tile per block (4 waves x 32)
constexpr int BN = 64;    // output columns per block (2 MFMA tiles per wave)
constexpr int KT = 32;    // reduction extent staged per step
constexpr int LDT = KT + 1;   // +1 float pad: fragment reads hit 32 distinct banks

// Stage a [rows x KT] tile of a row-major matrix into LDS as tile[r][kk] (kk contiguous).
// Out-of-range rows/cols are zero-filled so tails need no special MFMA handling.
__device__ __forceinline__ void stage_rows(float* __restrict__ tile, const float* __restrict__ G,
                                           int64_t ld, int64_t row0, int64_t row_end, int k0,
                                           int k_end, int rows, bool vec4) {
  if (vec4) {
    // each thread moves float4s: rows*KT/4 of them
    for (int idx = threadIdx.x; idx < rows * (KT / 4); idx += blockDim.x) {
      const int r = idx / (KT / 4), c4 = (idx % (KT / 4)) * 4;
      const int64_t gr = row0 + r;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (gr < row_end && k0 + c4 < k_end)   // k_end - k0 is a multiple of 4 on this path
        v = *reinterpret_cast<const float4*>(G + gr * ld + k0 + c4);
      float* t = tile + r * LDT + c4;
      t[0] = v.x; t[1] = v.y; t[2] = v.z; t[3] = v.w;
    }
  } else {
    for (int idx = threadIdx.x; idx < rows * KT; idx += blockDim.x) {
      const int r = idx / KT, c = idx % KT;
      const int64_t gr = row0 + r;
      float v = 0.f;
      if (gr < row_end && k0 + c < k_end) v = G[gr * ld + k0 + c];
      tile[r * LDT + c] = v;
    }
  }
}

// Stage B given as [Kred x Nout] row-major (bwd_input: W[kk][k0+j]) into tile[j][kk].
__device__ __forceinline__ void stage_cols(float* __restrict__ tile, const float* __restrict__ G,
                                           int64_t ld, int kk0, int kk_end, int j0, int j_end,
                                           int cols) {
  for (int idx = threadIdx.x; idx < cols * KT; idx += blockDim.x) {
    const int kk = idx / cols, j = idx % cols;     // consecutive threads -> consecutive j
    float v = 0.f;
    if (kk0 + kk < kk_end && j0 + j < j_end) v = G[(int64_t)(kk0 + kk) * ld + j0 + j];
    tile[j * LDT + kk] = v;
  }
}

__device__ __forceinline__ void mma_stage(const float* __restrict__ As, const float* __restrict__ Bs,
                                          int wave, int lane, int ksteps, f32x16& acc0,
                                          f32x16& acc1) {
  const float* a = As + (wave * 32 + (lane & 31)) * LDT + (lane >> 5);
  const float* b0 = Bs + (lane & 31) * LDT + (lane >> 5);
  const float* b1 = b0 + 32 * LDT;
#pragma unroll 4
  for (int s = 0; s < ksteps; ++s) {
    const float av = a[2 * s];
    const float bv0 = b0[2 * s];
    const float bv1 = b1[2 * s];
    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv0, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv1, acc1, 0, 0, 0);
  }
}

// ---------------------------------------------------------------------------- forward (NT)
__global__ void __launch_bounds__(256) k_linear_fwd(
    const float* __restrict__ X1, int64_t ldx1, int K1, const float* __restrict__ X2,
    int64_t ldx2, int K2, const float* __restrict__ W, const float* __restrict__ bias, int relu,
    float* __restrict__ Y, int64_t ldy, int64_t M, int N) {
  __shared__ float As[BM * LDT];
  __shared__ float Bs[BN * LDT];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t m0 = (int64_t)blockIdx.x * BM;
  const int n0 = blockIdx.y * BN;
  const int ldw = K1 + K2;
  f32x16 acc0 = {0}, acc1 = {0};
  for (int p = 0; p < 2; ++p) {
    const float* X = p == 0 ? X1 : X2;
    const int64_t ldx = p == 0 ? ldx1 : ldx2;
    const int K = p == 0 ? K1 : K2;
    const int koff = p == 0 ? 0 : K1;
    if (K == 0) continue;
    const bool vx = (K % 4 == 0) && (ldx % 4 == 0) && ((reinterpret_cast<uintptr_t>(X) & 15) == 0);
    const bool vw = (K % 4 == 0) && (ldw % 4 == 0) && (koff % 4 == 0) &&
                    ((reinterpret_cast<uintptr_t>(W) & 15) == 0);
    for (int k0 = 0; k0 < K; k0 += KT) {
      __syncthreads();
      stage_rows(As, X, ldx, m0, M, k0, K, BM, vx);
      stage_rows(Bs, W + koff, ldw, n0, N, k0, K, BN, vw);
      __syncthreads();
      const int ksteps = (min(KT, K - k0) + 1) / 2;
      mma_stage(As, Bs, wave, lane, ksteps, acc0, acc1);
    }
  }
  // epilogue
  const int j = lane & 31;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const f32x16& acc = t == 0 ? acc0 : acc1;
    const int col = n0 + t * 32 + j;
    if (col >= N) continue;
    const float bv = bias ? bias[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int64_t row = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (row < M) {
        float v = acc[r] + bv;
        if (relu) v = fmaxf(v, 0.f);
        Y[row * ldy + col] = v;
      }
    }
  }
}

// ----------------------------------------------------------------------- backward input (NN)
// dX[M, K] = dY[M, N] * W[N, k0:k0+K]; reduction over N.
__global__ void __launch_bounds__(256) k_linear_bwd_input(
    const float* __restrict__ dY, int64_t lddy, const float* __restrict__ W, int ldw, int kcol0,
    float* __restrict__ dX, int64_t lddx, int64_t M, int N, int K) {
  __shared__ float As[BM * LDT];
  __shared__ float Bs[BN * LDT];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t m0 = (int64_t)blockIdx.x * BM;
  const int j0 = blockIdx.y * BN;            // output column (k index of W) tile
  const bool va = (N % 4 == 0) && (lddy % 4 == 0) && ((reinterpret_cast<uintptr_t>(dY) & 15) == 0);
  f32x16 acc0 = {0}, acc1 = {0};
  for (int n0 = 0; n0 < N; n0 += KT) {
    __syncthreads();
    stage_rows(As, dY, lddy, m0, M, n0, N, BM, va);
    stage_cols(Bs, W + kcol0, ldw, n0, N, j0, K, BN);
    __syncthreads();
    const int ksteps = (min(KT, N - n0) + 1) / 2;
    mma_stage(As, Bs, wave, lane, ksteps, acc0, acc1);
  }
  const int j = lane & 31;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const f32x16& acc = t == 0 ? acc0 : acc1;
    const int col = j0 + t * 32 + j;
    if (col >= K) continue;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int64_t row = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (row < M) dX[row * lddx + col] = acc[r];
    }
  }
}

// ---------------------------------------------------------------------- backward weight (TN)
// partial[chunk][n][k] = sum_{m in chunk} dY[m][n] * X[m][k]
// Block: 4 waves, output tile 64(n) x 64(k); wave w owns the 32x32 tile (w>>1, w&1).
constexpr int MT = 32;           // rows of M staged per step
constexpr int WCHUNK = 512;      // rows of M per block (many blocks: M is the only big dim)
constexpr int LDM = 64 + 4;      // row stride of the staged [MT][64] tiles (16-B aligned rows)

__global__ void __launch_bounds__(256) k_linear_bwd_weight(
    const float* __restrict__ dY, int64_t lddy, const float* __restrict__ X, int64_t ldx,
    float* __restrict__ slab, int64_t M, int N, int K, int tiles_n, int tiles_k) {
  __shared__ float Ds[MT * LDM];
  __shared__ float Xs[MT * LDM];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tile = blockIdx.y;
  const int n0 = (tile / tiles_k) * 64, k0 = (tile % tiles_k) * 64;
  const int64_t mbeg = (int64_t)blockIdx.x * WCHUNK;
  const int64_t mend = min(M, mbeg + WCHUNK);
  const int tn = wave >> 1, tk = wave & 1;
  f32x16 acc = {0};
  for (int64_t m0 = mbeg; m0 < mend; m0 += MT) {
    __syncthreads();
    for (int idx = threadIdx.x; idx < MT * 64; idx += 256) {
      const int r = idx >> 6, c = idx & 63;
      const int64_t gr = m0 + r;
      float dv = 0.f, xv = 0.f;
      if (gr < mend) {
        if (n0 + c < N) dv = dY[gr * lddy + n0 + c];
        if (k0 + c < K) xv = X[gr * ldx + k0 + c];
      }
      Ds[r * LDM + c] = dv;
      Xs[r * LDM + c] = xv;
    }
    __syncthreads();
    const float* a = Ds + (lane >> 5) * LDM + tn * 32 + (lane & 31);
    const float* b = Xs + (lane >> 5) * LDM + tk * 32 + (lane & 31);
#pragma unroll 8
    for (int s = 0; s < MT / 2; ++s)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2 * s * LDM], b[2 * s * LDM], acc, 0, 0, 0);
  }
  // slab layout: [chunk][N][K]
  float* out = slab + (int64_t)blockIdx.x * N * K;
  const int col = k0 + tk * 32 + (lane & 31);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = n0 + tn * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    if (row < N && col < K) out[(int64_t)row * K + col] = acc[r];
  }
}

// 16 threads per output element walk the chunk dimension (fixed assignment -> deterministic),
// then a fixed-order shuffle tree; coalesced across elements.
__global__ void __launch_bounds__(256) k_reduce_slab(const float* __restrict__ slab,
                                                     int64_t nchunks, int64_t elems,
                                                     float* __restrict__ dW, int ldw, int kcol0,
                                                     int K) {
  const int64_t i = (int64_t)blockIdx.x * 16 + (threadIdx.x & 15);
  const int part = threadIdx.x >> 4;           // 0..15
  double s = 0.0;
  if (i < elems)
    for (int64_t c = part; c < nchunks; c += 16) s += (double)slab[c * elems + i];
  __shared__ double red[256];
  red[threadIdx.x] = s;
  __syncthreads();
  if (part == 0 && i < elems) {
    double t = 0.0;
#pragma unroll
    for (int p2 = 0; p2 < 16; ++p2) t += red[p2 * 16 + (threadIdx.x & 15)];
    const int64_t n = i / K, k = i % K;
    dW[n * ldw + kcol0 + k] = (float)t;
  }
}

// ------------------------------------------------------------------------------- column sums
constexpr int CS_ROWS = 512;     // rows per block

__global__ void __launch_bounds__(256) k_colsum_partial(const float* __restrict__ A, int64_t lda,
                                                        int64_t M, int N,
                                                        double* __restrict__ slab) {
  __shared__ double red[256];
  const int64_t rbeg = (int64_t)blockIdx.x * CS_ROWS;
  const int64_t rend = min(M, rbeg + CS_ROWS);
  for (int c0 = 0; c0 < N; c0 += 256) {
    const int nc = min(256, N - c0);
    const int rpi = 256 / nc;                 // rows handled per iteration
    const int c = threadIdx.x % nc, rr = threadIdx.x / nc;
    float sf = 0.f;                            // <= CS_ROWS/rpi terms per thread
    if (rr < rpi) {
      int64_t r = rbeg + rr;
      for (; r + 3 * rpi < rend; r += 4 * rpi) {   // 4 independent loads in flight
        const float v0 = A[r * lda + c0 + c], v1 = A[(r + rpi) * lda + c0 + c];
        const float v2 = A[(r + 2 * rpi) * lda + c0 + c], v3 = A[(r + 3 * rpi) * lda + c0 + c];
        sf += (v0 + v1) + (v2 + v3);
      }
      for (; r < rend; r += rpi) sf += A[r * lda + c0 + c];
    }
    red[threadIdx.x] = (double)sf;
    __syncthreads();
    if (threadIdx.x < nc) {
      double t = 0.0;
      for (int q = 0; q < rpi; ++q) t += red[q * nc + threadIdx.x];
      slab[(int64_t)blockIdx.x * N + c0 + threadIdx.x] = t;
    }
    __syncthreads();
  }
}

__global__ void __launch_bounds__(256) k_colsum_final(const double* __restrict__ slab,
                                                      int64_t nblocks, int N,
                                                      float* __restrict__ out) {
  const int c = blockIdx.x * 16 + (threadIdx.x & 15);
  const int part = threadIdx.x >> 4;
  double s = 0.0;
  if (c < N)
    for (int64_t b = part; b < nblocks; b += 16) s += slab[b * N + c];
  __shared__ double red[256];
  red[threadIdx.x] = s;
  __syncthreads();
  if (part == 0 && c < N) {
    double t = 0.0;
#pragma unroll
    for (int p2 = 0; p2 < 16; ++p2) t += red[p2 * 16 + (threadIdx.x & 15)];
    out[c] = (float)t;
  }
}

}  // namespace

extern "C" {

int cgnn_linear_fwd_f32(const float* X1, int64_t ldx1, int32_t K1, const float* X2, int64_t ldx2,
                        int32_t K2, const float* W, const float* bias, int32_t relu, float* Y,
                        int64_t ldy, int64_t M, int32_t N, void* stream) {
  if (M < 0 || N <= 0 || K1 <= 0 || K2 < 0 || ldx1 < K1 || ldy < N) return CGNN_EINVAL;
  if (K2 > 0 && (!X2 || ldx2 < K2)) return CGNN_EINVAL;
  if (M == 0) return CGNN_OK;
  if (!X1 || !W || !Y) return CGNN_EINVAL;
  if (cgnn_ws_linear_fwd(X1, ldx1, K1, X2, ldx2, K2, W, bias, relu, Y, ldy, M, N, nullptr,
                         cgnn_stream(stream))) {
    CGNN_CHECK_LAUNCH();
    return CGNN_OK;
  }
  if (N == 256 && M >= 4096 && (int64_t)128 * (K1 + K2) <= 32768) {
    // two 128-column halves, each with its weight panel parked in LDS (gemm_ws.hip)
    const int K = K1 + K2;
    if (cgnn_ws_linear_fwd(X1, ldx1, K1, X2, ldx2, K2, W, bias, relu, Y, ldy, M, 128, nullptr,
                           cgnn_stream(stream)) &&
        cgnn_ws_linear_fwd(X1, ldx1, K1, X2, ldx2, K2, W + (int64_t)128 * K, bias ? bias + 128 : nullptr,
                           relu, Y + 128, ldy, M, 128, nullptr, cgnn_stream(stream))) {
      CGNN_CHECK_LAUNCH();
      return CGNN_OK;
    }
  }
  dim3 grid((unsigned)((M + BM - 1) / BM), (unsigned)((N + BN - 1) / BN));
  k_linear_fwd<<<grid, 256, 0, cgnn_stream(stream)>>>(X1, ldx1, K1, X2, ldx2, K2, W, bias, relu, Y,
                                                      ldy, M, N);
  CGNN_CHECK_LAUNCH();
  return CGNN_OK;
}

int cgnn_linear_fwd_stats_f32(const float* X1, int64_t ldx1, int32_t K1, const float* X2,
                              int64_t ldx2, int32_t K2, const float* W, const float* bias,
                              int32_t relu, float* Y, int64_t ldy, int64_t M, int32_t N,
                              double* stat_slab, int64_t stat_slab_bytes, void* stream) {
  if (M <= 0 || N <= 0 || K1 <= 0 || K2 < 0 || ldx1 < K1 || ldy < N || !stat_slab) return CGNN_EINVAL;
  if (K2 > 0 && (!X2 || ldx2 < K2)) return CGNN_EINVAL;
  if (!X1 || !W || !Y) return CGNN_EINVAL;
  CGNN_NEED_BYTES(stat_slab, stat_slab_bytes, (int64_t)cgnn_fused_grid() * 2 * N * (int64_t)sizeof(double));
  if (!cgnn_ws_linear_fwd(X1, ldx1, K1, X2, ldx2, K2, W, bias, relu, Y, ldy, M, N, stat_slab,
                          cgnn_stream(stream)))
    return CGNN_EUNSUPPORTED;      // caller: cgnn_linear_fwd_f32 + cgnn_bn_act_fwd_stats
  CGNN_CHECK_LAUNCH();
  return CGNN_OK;
}

int cgnn_linear_bwd_input_f32(const float* dY, int64_t lddy, const float* W, int32_t ldw,
                              int32_t k0, float* dX, int64_t lddx, int64_t M, int32_t N, int32_t K,
                              void* stream) {
  if (M < 0 || N <= 0 || K <= 0 || k0 < 0 || ldw < k0 + K || lddy < N || lddx < K)
    return CGNN_EINVAL;
  if (M == 0) return CGNN_OK;
  if (!dY || !W || !dX) return CGNN_EINVAL;
  if (cgnn_ws_linear_bwd_input(dY, lddy, W, ldw, k0, dX, lddx, M, N, K, cgnn_stream(stream))) {
    CGNN_CHECK_LAUNCH();
    return CGNN_OK;
  }
  if (K == 256 && M >= 4096 && (int64_t)N * 128 <= 32768) {
    // two halves of 128 output columns, each with its [N x 128] weight panel in LDS
    if (cgnn_ws_linear_bwd_input(dY, lddy, W, ldw, k0, dX, lddx, M, N, 128, cgnn_stream(stream)) &&
        cgnn_ws_linear_bwd_input(dY, lddy, W, ldw, k0 + 128, dX + 128, lddx, M, N, 128,
                                 cgnn_stream(stream))) {
      CGNN_CHECK_LAUNCH();
      return CGNN_OK;
    }
  }
  dim3 grid((unsigned)((M + BM - 1) / BM), (unsigned)((K + BN - 1) / BN));
  k_linear_bwd_input<<<grid, 256, 0, cgnn_stream(stream)>>>(dY, lddy, W, ldw, k0, dX, lddx, M, N, K);
  CGNN_CHECK_LAUNCH();
  return CGNN_OK;
}

// fp32 words one cgnn_linear_bwd_weight_f32(M, N, K) call writes into its slab: the generic kernel's row
// chunks, or the weight-stationary kernel's one partial per workgroup -- for N = 256 two halves of 128
// rows, one at a time
static int64_t bwd_weight_slab_words(int64_t M, int32_t N, int32_t K) {
  int64_t nchunks = (M + WCHUNK - 1) / WCHUNK;
  if (nchunks == 0) nchunks = 1;
  int64_t words = nchunks * (int64_t)N * K;
  const int64_t ws = cgnn_ws_bwd_weight_partials(M, N, K) * (int64_t)N * K;
  if (ws > words) words = ws;
  if (N == 256) {
    const int64_t half = cgnn_ws_bwd_weight_partials(M, 128, K) * (int64_t)128 * K;
    if (half > words) words = half;
  }
  return words;
}

int64_t cgnn_linear_bwd_weight_workspace_bytes(int64_t M, int32_t N, int32_t K) {
  if (M < 0 || N <= 0 || K <= 0) return CGNN_EINVAL;
  return cgnn_align_up(bwd_weight_slab_words(M, N, K) * (int64_t)sizeof(float), 256);
}

// both panels in one pass over dY when the joint shape fits the weight-stationary kernel, else one
// cgnn_linear_bwd_weight_f32 per panel, each of which may take ITS weight-stationary form (one partial of
// [N x Ki] per workgroup -- GraphSAGE hidden 256 on >= 4096 nodes wrote 33 MB into a slab sized for the
// joint generic form): the slab must hold the largest of the three
static int64_t bwd_weight2_slab_words(int64_t M, int32_t N, int32_t K1, int32_t K2) {
  int64_t words = cgnn_ws_bwd_weight_partials(M, N, K1 + K2) * (int64_t)N * (K1 + K2);
  const int64_t w1 = bwd_weight_slab_words(M, N, K1), w2 = bwd_weight_slab_words(M, N, K2);
  if (w1 > words) words = w1;
  if (w2 > words) words = w2;
  return words;
}

int64_t cgnn_linear_bwd_weight2_workspace_bytes(int64_t M, int32_t N, int32_t K1, int32_t K2) {
  if (M < 0 || N <= 0 || K1 <= 0 || K2 <= 0) return CGNN_EINVAL;
  return cgnn_align_up(bwd_weight2_slab_words(M, N, K1, K2) * (int64_t)sizeof(float), 256);
}

int cgnn_linear_bwd_weight_f32(const float* dY, int64_t lddy, const float* X, int64_t ldx,
                               float* dW, int32_t ldw, int32_t k0, int64_t M, int32_t N, int32_t K,
                               void* slab, int64_t slab_bytes, void* stream) {
  if (M < 0 || N <= 0 || K <= 0 || k0 < 0 || ldw < k0 + K || lddy < N || ldx < K)
    return CGNN_EINVAL;
  if (!dW || !slab) return CGNN_EINVAL;
  if (M > 0 && (!dY || !X)) return CGNN_EINVAL;
  CGNN_NEED_BYTES(slab, slab_bytes, bwd_weight_slab_words(M, N, K) * (int64_t)sizeof(float));
  hipStream_t st = cgnn_stream(stream);
  if (N == 256 && cgnn_ws_bwd_weight_partials(M, 128, K) > 0 && lddy % 4 == 0) {
    // two halves of 128 output rows of dW through the weight-stationary kernel
    const int rc = cgnn_linear_bwd_weight_f32(dY, lddy, X, ldx, dW, ldw, k0, M, 128, K, slab, slab_bytes, stream);
    if (rc != CGNN_OK) return rc;
    return cgnn_linear_bwd_weight_f32(dY + 128, lddy, X, ldx, dW + (int64_t)128 * ldw, ldw, k0, M, 128, K,
                                      slab, slab_bytes, stream);
  }
  int64_t nchunks = (M + WCHUNK - 1) / WCHUNK;
  const int tiles_n = (N + 63) / 64, tiles_k = (K + 63) / 64;
  if (cgnn_ws_linear_bwd_weight(dY, lddy, X, ldx, K, nullptr, 0, 0, static_cast<float*>(slab), M, N, st)) {
    CGNN_CHECK_LAUNCH();
    nchunks = cgnn_ws_bwd_weight_partials(M, N, K);
  } else if (nchunks > 0) {
    dim3 grid((unsigned)nchunks, (unsigned)(tiles_n * tiles_k));
    k_linear_bwd_weight<<<grid, 256, 0, st>>>(dY, lddy, X, ldx, static_cast<float*>(slab), M, N, K,
                                              tiles_n, tiles_k);
    CGNN_CHECK_LAUNCH();
  }
  const int64_t elems = (int64_t)N * K;
  k_reduce_slab<<<(unsigned)((elems + 15) / 16), 256, 0, st>>>(static_cast<float*>(slab), nchunks,
                                                                 elems, dW, ldw, k0, K);
  CGNN_CHECK_LAUNCH();
  return CGNN_OK;
}

int cgnn_linear_bwd_weight2_f32(const float* dY, int64_t lddy, const float* X1, int64_t ldx1,
                                int32_t K1, const float* X2, int64_t ldx2, int32_t K2, float* dW,
                                int32_t ldw, int64_t M, int32_t N, void* slab, int64_t slab_bytes, void* stream) {
  if (M < 0 || N <= 0 || K1 <= 0 || K2 <= 0 || ldw < K1 + K2 || lddy < N || ldx1 < K1 || ldx2 < K2)
    return CGNN_EINVAL;
  if (!dW || !slab) return CGNN_EINVAL;
  if (M > 0 && (!dY || !X1 || !X2)) return CGNN_EINVAL;
  CGNN_NEED_BYTES(slab, slab_bytes, bwd_weight2_slab_words(M, N, K1, K2) * (int64_t)sizeof(float));
  hipStream_t st = cgnn_stream(stream);
  if (cgnn_ws_linear_bwd_weight(dY, lddy, X1, ldx1, K1, X2, ldx2, K2, static_cast<float*>(slab), M, N, st)) {
    CGNN_CHECK_LAUNCH();
    const int64_t elems = (int64_t)N * (K1 + K2);
    k_reduce_slab<<<(unsigned)((elems + 15) / 16), 256, 0, st>>>(
        static_cast<float*>(slab), cgnn_ws_bwd_weight_partials(M, N, K1 + K2), elems, dW, ldw, 0, K1 + K2);
    CGNN_CHECK_LAUNCH();
    return CGNN_OK;
  }
  // shapes outside the weight-stationary kernel: one panel at a time
  const int rc = cgnn_linear_bwd_weight_f32(dY, lddy, X1, ldx1, dW, ldw, 0, M, N, K1, slab, slab_bytes, stream);
  if (rc != CGNN_OK) return rc;
  return cgnn_linear_bwd_weight_f32(dY, lddy, X2, ldx2, dW, ldw, K1, M, N, K2, slab, slab_bytes, stream);
}

int64_t cgnn_colsum_workspace_bytes(int64_t M, int32_t N) {
  if (M < 0 || N <= 0) return CGNN_EINVAL;
  int64_t nb = (M + CS_ROWS - 1) / CS_ROWS;
  if (nb == 0) nb = 1;
  return cgnn_align_up(nb * (int64_t)N * (int64_t)sizeof(double), 256);
}

int cgnn_colsum_f32(const float* A, int64_t lda, float* out, int64_t M, int32_t N, void* slab, int64_t slab_bytes,
                    void* stream) {
  if (M < 0 || N <= 0 || lda < N || !out || !slab) return CGNN_EINVAL;
  if (M > 0 && !A) return CGNN_EINVAL;
  hipStream_t st = cgnn_stream(stream);
  const int64_t nb = (M + CS_ROWS - 1) / CS_ROWS;
  CGNN_NEED_BYTES(slab, slab_bytes, (nb > 0 ? nb : 1) * (int64_t)N * (int64_t)sizeof(double));
  if (nb > 0) {
    k_colsum_partial<<<(unsigned)nb, 256, 0, st>>>(A, lda, M, N, static_cast<double*>(slab));
    CGNN_CHECK_LAUNCH();
  }
  k_colsum_final<<<(N + 15) / 16, 256, 0, st>>>(static_cast<double*>(slab), nb, N, out);
  CGNN_CHECK_LAUNCH();
  return CGNN_OK;
}

}  // extern "C"
