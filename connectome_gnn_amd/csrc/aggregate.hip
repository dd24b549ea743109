// aggregate.hip -- edge-weighted segment reduction over a CSR (no atomics).
//
//   Y[r,:] = ( sum_{s in row r} coef[s] * X[col[s],:] ) / rowdiv[r] + selfc[r]*X[r,:] + bias
//
// Replaces gather -> mul -> scatter_add_ of models.py:112-114 (GCN) and :146-149 (SAGE), and,
// on the src-sorted CSR, autograd's index_put_(accumulate) backward of x[src].
//
// Generic (any graph size, any edge pattern) form: one wave per destination row, the wave's
// 64 lanes span the feature columns (VEC consecutive floats per lane), neighbour rows are
// gathered from L2/HBM as whole coalesced 256*VEC-byte lines.  Slot metadata (col, coef) is
// loaded 64 slots at a time, one per lane, and broadcast with v_readlane.  Slots are summed in
// order, so the result is run-to-run deterministic and follows the reference's COO order.
// The per-graph LDS-staged form used by the fused layer kernels lives in fused_gcn.hip.
#include "common.h"

namespace {

template <int VEC> struct VecT;
template <> struct VecT<1> { using type = float; };
template <> struct VecT<2> { using type = float2; };
template <> struct VecT<4> { using type = float4; };

template <int VEC>
__device__ __forceinline__ void fma_vec(float (&acc)[VEC], float c,
                                        const typename VecT<VEC>::type& v) {
  const float* p = reinterpret_cast<const float*>(&v);
#pragma unroll
  for (int i = 0; i < VEC; ++i) acc[i] = fmaf(c, p[i], acc[i]);
}

// F == 64*VEC.  Block = 4 waves = 4 rows.  ACC: the row's sum is added onto what Y holds (the dense
// fragments' part, written by band_aggregate.hip just before); that read is issued with the row's
// first gathers, so it costs one more line among the ~16 in flight.
template <int VEC, bool ACC = false>
__global__ void __launch_bounds__(256) k_agg_wave_row(
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
    const float* __restrict__ coef, const float* __restrict__ selfc,
    const float* __restrict__ rowdiv, const float* __restrict__ bias,
    const float* __restrict__ X, int64_t ldx, float* __restrict__ Y, int64_t ldy,
    int64_t num_rows) {
  using V = typename VecT<VEC>::type;
  const int lane = threadIdx.x & 63;
  // XCD-aware row order: workgroup b runs on XCD b % 8 (its own L2), so XCD x walks the x-th
  // EIGHTH of the rows front to back.  With rows 4b.. on workgroup b (round 2) every XCD touched
  // every graph, and each of the eight L2s pulled all of X from HBM: PMC 1.70 GB per launch against
  // 0.26 GB of compulsory bytes at 64 x 1000-ROI, F = 256 (a graph's [1000 x 256] block is 1 MB --
  // it fits one L2 and is re-read ~100 times by its own rows).
  const int xcd = blockIdx.x & 7;
  const int64_t nrb = (num_rows + 3) / 4, per = (nrb + 7) / 8;
  for (int64_t i = blockIdx.x >> 3; i < per; i += gridDim.x >> 3) {
    const int64_t r = (xcd * per + i) * 4 + (threadIdx.x >> 6);
    if (r >= num_rows) continue;
    const int beg = cgnn_uniform(rowptr[r]);
    const int end = cgnn_uniform(rowptr[r + 1]);
    float acc[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
    V yold;
    if (ACC) yold = *reinterpret_cast<const V*>(Y + r * ldy + lane * VEC);
    for (int base = beg; base < end; base += 64) {
      const int cnt = min(64, end - base);
      int mycol = 0;
      float mycoef = 0.f;
      if (lane < cnt) {
        mycol = col[base + lane];
        mycoef = coef[base + lane];
      }
      int j = 0;
      for (; j + 4 <= cnt; j += 4) {
        int c0 = __builtin_amdgcn_readlane(mycol, j), c1 = __builtin_amdgcn_readlane(mycol, j + 1);
        int c2 = __builtin_amdgcn_readlane(mycol, j + 2), c3 = __builtin_amdgcn_readlane(mycol, j + 3);
        V v0 = *reinterpret_cast<const V*>(X + (int64_t)c0 * ldx + lane * VEC);
        V v1 = *reinterpret_cast<const V*>(X + (int64_t)c1 * ldx + lane * VEC);
        V v2 = *reinterpret_cast<const V*>(X + (int64_t)c2 * ldx + lane * VEC);
        V v3 = *reinterpret_cast<const V*>(X + (int64_t)c3 * ldx + lane * VEC);
        float w0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mycoef), j));
        float w1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mycoef), j + 1));
        float w2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mycoef), j + 2));
        float w3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mycoef), j + 3));
        fma_vec<VEC>(acc, w0, v0);
        fma_vec<VEC>(acc, w1, v1);
        fma_vec<VEC>(acc, w2, v2);
        fma_vec<VEC>(acc, w3, v3);
      }
      for (; j < cnt; ++j) {
        int c0 = __builtin_amdgcn_readlane(mycol, j);
        float w0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mycoef), j));
        V v0 = *reinterpret_cast<const V*>(X + (int64_t)c0 * ldx + lane * VEC);
        fma_vec<VEC>(acc, w0, v0);
      }
    }
    if (rowdiv) {
      const float dv = rowdiv[r];
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc[i] = acc[i] / dv;
    }
    if (selfc) {
      const float sc = selfc[r];
      V v = *reinterpret_cast<const V*>(X + r * ldx + lane * VEC);
      fma_vec<VEC>(acc, sc, v);                 // self-loop term last (models.py:98-100)
    }
    if (bias) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc[i] += bias[lane * VEC + i];
    }
    V o;
    float* po = reinterpret_cast<float*>(&o);
#pragma unroll
    for (int i = 0; i < VEC; ++i) po[i] = ACC ? reinterpret_cast<const float*>(&yold)[i] + acc[i] : acc[i];
    *reinterpret_cast<V*>(Y + r * ldy + lane * VEC) = o;
  }
}

// Any F: one thread per (row, column).  Used for the narrow input layer (F = 5, 10, 20, 32).
// (Measured and not kept: 16 lanes per row with a butterfly fold -- 46 us against 107 on [64000 x 5] --
// moved the 1000-ROI layer-0 weight gradient from 5e-7 to 4e-5 of the fp64 oracle: that gradient is a
// cancellation against the BatchNorm mean and the CSR-order FMA chain is what keeps it tight.)
__global__ void __launch_bounds__(256) k_agg_elem(
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
    const float* __restrict__ coef, const float* __restrict__ selfc,
    const float* __restrict__ rowdiv, const float* __restrict__ bias,
    const float* __restrict__ X, int64_t ldx, float* __restrict__ Y, int64_t ldy,
    int64_t num_rows, int F) {
  const int64_t total = num_rows * F;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = t / F;
    const int c = (int)(t - r * F);
    float acc = 0.f;
    int s = rowptr[r];
    const int e = rowptr[r + 1];
    for (; s + 8 <= e; s += 8) {                 // eight gathers in flight; the FMA chain stays in CSR order
      float cf[8], xv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        cf[u] = coef[s + u];
        xv[u] = X[(int64_t)col[s + u] * ldx + c];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) acc = fmaf(cf[u], xv[u], acc);
    }
    for (; s < e; ++s) acc = fmaf(coef[s], X[(int64_t)col[s] * ldx + c], acc);
    if (rowdiv) acc = acc / rowdiv[r];
    if (selfc) acc = fmaf(selfc[r], X[r * ldx + c], acc);
    if (bias) acc += bias[c];
    Y[r * ldy + c] = acc;
  }
}

}  // namespace

static bool agg_vec_ok(const float* X, int64_t ldx, const float* Y, int64_t ldy, int F) {
  return (F % 64 == 0) && (F / 64 == 1 || F / 64 == 2 || F / 64 == 4) && (ldx % (F / 64) == 0) &&
         (ldy % (F / 64) == 0) && (reinterpret_cast<uintptr_t>(X) % (4 * (F / 64)) == 0) &&
         (reinterpret_cast<uintptr_t>(Y) % (4 * (F / 64)) == 0);
}

template <bool ACC>
static void agg_wave_row_launch(const int32_t* rowptr, const int32_t* col, const float* coef, const float* selfc,
                                const float* rowdiv, const float* bias, const float* X, int64_t ldx, float* Y,
                                int64_t ldy, int64_t num_rows, int F, hipStream_t st) {
  const int64_t per = ((num_rows + 3) / 4 + 7) / 8;          // 4-row blocks per XCD
  unsigned grid = (unsigned)(8 * (per < 2048 ? per : 2048)); // a multiple of 8; the rest grid-strides
  switch (F / 64) {
    case 1: k_agg_wave_row<1, ACC><<<grid, 256, 0, st>>>(rowptr, col, coef, selfc, rowdiv, bias, X, ldx, Y, ldy, num_rows); break;
    case 2: k_agg_wave_row<2, ACC><<<grid, 256, 0, st>>>(rowptr, col, coef, selfc, rowdiv, bias, X, ldx, Y, ldy, num_rows); break;
    default: k_agg_wave_row<4, ACC><<<grid, 256, 0, st>>>(rowptr, col, coef, selfc, rowdiv, bias, X, ldx, Y, ldy, num_rows); break;
  }
}

extern "C" {

int cgnn_aggregate_f32(const int32_t* rowptr, const int32_t* col, const float* coef,
                       const float* selfc, const float* rowdiv, const float* bias,
                       const float* X, int64_t ldx, float* Y, int64_t ldy,
                       int64_t num_rows, int32_t F, void* stream) {
  if (num_rows < 0 || F <= 0 || ldx < F || ldy < F) return CGNN_EINVAL;
  if (num_rows == 0) return CGNN_OK;
  if (!rowptr || !X || !Y) return CGNN_EINVAL;
  hipStream_t st = cgnn_stream(stream);
  if (agg_vec_ok(X, ldx, Y, ldy, F)) {
    agg_wave_row_launch<false>(rowptr, col, coef, selfc, rowdiv, bias, X, ldx, Y, ldy, num_rows, F, st);
  } else {
    int64_t total = num_rows * F;
    unsigned grid = (unsigned)((total + 255) / 256);
    if (grid > 256u * 32u) grid = 256u * 32u;
    k_agg_elem<<<grid, 256, 0, st>>>(rowptr, col, coef, selfc, rowdiv, bias, X, ldx, Y, ldy, num_rows, F);
  }
  CGNN_CHECK_LAUNCH();
  return CGNN_OK;
}

int cgnn_aggregate_acc_f32(const int32_t* rowptr, const int32_t* col, const float* coef,
                           const float* selfc, const float* rowdiv, const float* bias,
                           const float* X, int64_t ldx, float* Y, int64_t ldy,
                           int64_t num_rows, int32_t F, void* stream) {
  if (num_rows < 0 || F <= 0 || ldx < F || ldy < F) return CGNN_EINVAL;
  if (!agg_vec_ok(X, ldx, Y, ldy, F)) return CGNN_EUNSUPPORTED;
  if (num_rows == 0) return CGNN_OK;
  if (!rowptr || !X || !Y || X == Y) return CGNN_EINVAL;
  agg_wave_row_launch<true>(rowptr, col, coef, selfc, rowdiv, bias, X, ldx, Y, ldy, num_rows, F, cgnn_stream(stream));
  CGNN_CHECK_LAUNCH();
  return CGNN_OK;
}

}  // extern "C"
