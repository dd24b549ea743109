// pool.hip -- per-graph mean-pool readout and its backward.
//
// Replaces _scatter_mean/_graph_mean_pool (models.py:40-47,57-59).  Nodes of a graph are
// contiguous in a ConnectomeBatch (graph.py:149-158), so the scatter over `batch` ids is a
// contiguous segment mean over [gptr[g], gptr[g+1]); the count is n_g and the divisor
// n_g + 1e-8 (models.py:47).
#include "common.h"

namespace {

// one block per (32-column group, graph): 8 row lanes x 32 columns; a row lane walks every 8th row with
// eight loads in flight, sums in double, and the 8 partials fold in a fixed order (reproducible).
// [64 graphs x 1000 rows x 256] was 271 us with one block per graph and one load in flight per thread.
__global__ void __launch_bounds__(256) k_pool_mean_fwd(const float* __restrict__ X, int64_t ldx,
                                                        const int32_t* __restrict__ gptr,
                                                        float* __restrict__ P, int F) {
  __shared__ double red[256];
  const int g = blockIdx.y, c = 32 * blockIdx.x + (threadIdx.x & 31), rr = threadIdx.x >> 5;
  const int rbeg = gptr[g], rend = gptr[g + 1];
  const float inv = 1.0f / ((float)(rend - rbeg) + 1e-8f);
  double s0 = 0.0, s1 = 0.0;
  if (c < F) {
    const float* xc = X + c;
    int r = rbeg + rr;
    for (; r + 56 < rend; r += 64) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = xc[(int64_t)(r + 8 * u) * ldx];
#pragma unroll
      for (int u = 0; u < 8; u += 2) {
        s0 += (double)v[u];
        s1 += (double)v[u + 1];
      }
    }
    for (; r < rend; r += 8) s0 += (double)xc[(int64_t)r * ldx];
  }
  red[threadIdx.x] = s0 + s1;
  __syncthreads();
  if (threadIdx.x < 32 && c < F) {
    double t = 0.0;
#pragma unroll
    for (int q = 0; q < 8; ++q) t += red[32 * q + threadIdx.x];
    P[(int64_t)g * F + c] = (float)t * inv;
  }
}

// grid (chunks, graphs): a graph's [n x F] block is cut into chunks of consecutive elements
__global__ void __launch_bounds__(256) k_pool_mean_bwd(const float* __restrict__ dP,
                                                        const int32_t* __restrict__ gptr,
                                                        float* __restrict__ dX, int64_t lddx,
                                                        int F) {
  const int g = blockIdx.y;
  const int rbeg = gptr[g], rend = gptr[g + 1];
  const float inv = 1.0f / ((float)(rend - rbeg) + 1e-8f);
  const int64_t total = (int64_t)(rend - rbeg) * F;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
    const int r = (int)(t / F), c = (int)(t - (int64_t)r * F);
    dX[(int64_t)(rbeg + r) * lddx + c] = dP[(int64_t)g * F + c] * inv;
  }
}

}  // namespace

extern "C" {

int cgnn_pool_mean_fwd_f32(const float* X, int64_t ldx, const int32_t* gptr, float* P,
                           int32_t num_graphs, int32_t F, void* stream) {
  if (num_graphs < 0 || F <= 0 || ldx < F) return CGNN_EINVAL;
  if (num_graphs == 0) return CGNN_OK;
  if (!X || !gptr || !P) return CGNN_EINVAL;
  // graphs ride on gridDim.y (<= 65535): larger batches (whole-dataset evaluation of small graphs) go in runs
  for (int32_t g0 = 0; g0 < num_graphs; g0 += 65535) {
    const int32_t ng = num_graphs - g0 < 65535 ? num_graphs - g0 : 65535;
    k_pool_mean_fwd<<<dim3((unsigned)((F + 31) / 32), (unsigned)ng), 256, 0, cgnn_stream(stream)>>>(
        X, ldx, gptr + g0, P + (int64_t)g0 * F, F);
    CGNN_CHECK_LAUNCH();
  }
  return CGNN_OK;
}

int cgnn_pool_mean_bwd_f32(const float* dP, const int32_t* gptr, float* dX, int64_t lddx,
                           int32_t num_graphs, int32_t F, void* stream) {
  if (num_graphs < 0 || F <= 0 || lddx < F) return CGNN_EINVAL;
  if (num_graphs == 0) return CGNN_OK;
  if (!dP || !gptr || !dX) return CGNN_EINVAL;
  // chunks per graph: enough blocks to fill the chip at any batch size, at most one per 1024 elements of a 1024-row graph
  const unsigned chunks = (unsigned)(num_graphs >= 2048 ? 1 : (2048 + num_graphs - 1) / num_graphs);
  for (int32_t g0 = 0; g0 < num_graphs; g0 += 65535) {        // (gridDim.y <= 65535, as in the forward)
    const int32_t ng = num_graphs - g0 < 65535 ? num_graphs - g0 : 65535;
    k_pool_mean_bwd<<<dim3(chunks, (unsigned)ng), 256, 0, cgnn_stream(stream)>>>(dP + (int64_t)g0 * F, gptr + g0, dX,
                                                                                lddx, F);
    CGNN_CHECK_LAUNCH();
  }
  return CGNN_OK;
}

}  // extern "C"
