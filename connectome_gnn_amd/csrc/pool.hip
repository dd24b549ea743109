// pool.hip -- per-graph mean-pool readout and its backward.
//
// Replaces _scatter_mean/_graph_mean_pool (models.py:40-47,57-59).  Nodes of a graph are
// contiguous in a ConnectomeBatch (graph.py:149-158), so the scatter over `batch` ids is a
// contiguous segment mean over [gptr[g], gptr[g+1]); the count is n_g and the divisor
// n_g + 1e-8 (models.py:47).
#include "common.h"

namespace {

__global__ void __launch_bounds__(256) k_pool_mean_fwd(const float* __restrict__ X, int64_t ldx,
                                                        const int32_t* __restrict__ gptr,
                                                        float* __restrict__ P, int F) {
  __shared__ double red[256];
  const int g = blockIdx.x;
  const int rbeg = gptr[g], rend = gptr[g + 1];
  const float inv = 1.0f / ((float)(rend - rbeg) + 1e-8f);
  for (int c0 = 0; c0 < F; c0 += 256) {
    const int nc = min(256, F - c0);
    const int rpi = 256 / nc;
    const int c = threadIdx.x % nc, rr = threadIdx.x / nc;
    double s = 0.0;
    if (rr < rpi)
      for (int r = rbeg + rr; r < rend; r += rpi) s += (double)X[(int64_t)r * ldx + c0 + c];
    red[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x < nc) {
      double t = 0.0;
      for (int q = 0; q < rpi; ++q) t += red[q * nc + threadIdx.x];
      P[(int64_t)g * F + c0 + threadIdx.x] = (float)t * inv;
    }
    __syncthreads();
  }
}

__global__ void __launch_bounds__(256) k_pool_mean_bwd(const float* __restrict__ dP,
                                                        const int32_t* __restrict__ gptr,
                                                        float* __restrict__ dX, int64_t lddx,
                                                        int F) {
  const int g = blockIdx.x;
  const int rbeg = gptr[g], rend = gptr[g + 1];
  const float inv = 1.0f / ((float)(rend - rbeg) + 1e-8f);
  const int64_t total = (int64_t)(rend - rbeg) * F;
  for (int64_t t = threadIdx.x; t < total; t += 256) {
    const int r = (int)(t / F), c = (int)(t % F);
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
  k_pool_mean_fwd<<<num_graphs, 256, 0, cgnn_stream(stream)>>>(X, ldx, gptr, P, F);
  CGNN_CHECK_LAUNCH();
  return CGNN_OK;
}

int cgnn_pool_mean_bwd_f32(const float* dP, const int32_t* gptr, float* dX, int64_t lddx,
                           int32_t num_graphs, int32_t F, void* stream) {
  if (num_graphs < 0 || F <= 0 || lddx < F) return CGNN_EINVAL;
  if (num_graphs == 0) return CGNN_OK;
  if (!dP || !gptr || !dX) return CGNN_EINVAL;
  k_pool_mean_bwd<<<num_graphs, 256, 0, cgnn_stream(stream)>>>(dP, gptr, dX, lddx, F);
  CGNN_CHECK_LAUNCH();
  return CGNN_OK;
}

}  // extern "C"
