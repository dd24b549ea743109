// adam.hip -- the optimizer update of a training step as ONE launch (gfx950).
//
// The reference trains with torch.optim.Adam(lr, weight_decay) (train.py / demo.py:105-134); torch's
// fused multi-tensor Adam is two launches (step counters += 1, then the update) and takes 7-43 us
// for the 15 k - 400 k parameters of these models.  Here every parameter tensor of a group is
// updated by one grid (1024 elements per workgroup), and the shared step counter is advanced by
// the same launch: every workgroup reads the counter, announces that it has (one atomic), and the
// last one to arrive writes counter + 1 -- nothing else of the launch reads it afterwards.
//
// Arithmetic = torch's Adam, L2 weight decay (not decoupled), no amsgrad:
//   g += wd * p;  m = b1 m + (1 - b1) g;  v = b2 v + (1 - b2) g^2
//   p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
#include "common.h"

namespace {

constexpr int ADAM_THR = 256;
constexpr int ADAM_PER_BLOCK = 4 * ADAM_THR;

__global__ void __launch_bounds__(ADAM_THR) k_adam(cgnn_adam_jobs jobs, float* __restrict__ step,
                                                   uint32_t* __restrict__ arrivals, int advance,
                                                   double lr_d, double b1_d, double b2_d, float eps, float wd) {
  __shared__ float s_step;
  if (threadIdx.x == 0) {
    const float s = __uint_as_float(__atomic_load_n(reinterpret_cast<uint32_t*>(step), __ATOMIC_RELAXED));
    s_step = s;
    if (advance) {
      __threadfence();                              // the read above is complete before we announce it
      const uint32_t old = atomicAdd(arrivals, 1u);
      if (old == gridDim.x - 1) {                   // every workgroup has read `step`
        __atomic_store_n(reinterpret_cast<uint32_t*>(step), __float_as_uint(s + 1.0f), __ATOMIC_RELAXED);
        __atomic_store_n(arrivals, 0u, __ATOMIC_RELAXED);
      }
    }
  }
  __syncthreads();
  // hyper-parameters arrive as the host's doubles and are rounded where torch rounds them:
  // beta -> float for the multiply, (1 - beta) computed in double THEN rounded (1 - 0.999f is
  // 1.3e-5 off 0.001f), the bias corrections and lr / bc1 in double
  const double t = (double)s_step + 1.0;
  const float b2 = (float)b2_d, w1 = (float)(1.0 - b1_d), w2 = (float)(1.0 - b2_d);
  const float bc2_sqrt = (float)sqrt(1.0 - pow(b2_d, t));
  const float step_size = (float)(lr_d / (1.0 - pow(b1_d, t)));

  int block = blockIdx.x;
#pragma unroll 1
  for (int i = 0; i < jobs.n; ++i) {
    const int64_t n = jobs.numel[i];
    const int nb = (int)((n + ADAM_PER_BLOCK - 1) / ADAM_PER_BLOCK);
    if (block < nb) {
      float* __restrict__ p = jobs.param[i];
      const float* __restrict__ g = jobs.grad[i];
      float* __restrict__ m = jobs.exp_avg[i];
      float* __restrict__ v = jobs.exp_avg_sq[i];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int64_t e = (int64_t)block * ADAM_PER_BLOCK + u * ADAM_THR + threadIdx.x;
        if (e < n) {
          const float pe = p[e];
          const float ge = fmaf(wd, pe, g[e]);
          const float m0 = m[e];
          const float me = fmaf(w1, ge - m0, m0);                 // lerp(m, g, 1 - b1)
          const float ve = fmaf(w2 * ge, ge, b2 * v[e]);
          m[e] = me;
          v[e] = ve;
          p[e] = pe - step_size * me / (sqrtf(ve) / bc2_sqrt + eps);
        }
      }
      return;
    }
    block -= nb;
  }
}

}  // namespace

extern "C" int cgnn_adam_step(const cgnn_adam_jobs* jobs, float* step, uint32_t* arrivals,
                              int32_t advance, double lr, double beta1, double beta2, double eps,
                              double weight_decay, void* stream) {
  if (!jobs || jobs->n < 1 || jobs->n > CGNN_ADAM_MAX_JOBS || !step || !arrivals) return CGNN_EINVAL;
  if (!(lr >= 0.0) || !(beta1 >= 0.0 && beta1 < 1.0) || !(beta2 >= 0.0 && beta2 < 1.0) || !(eps >= 0.0))
    return CGNN_EINVAL;
  int64_t blocks = 0;
  for (int i = 0; i < jobs->n; ++i) {
    if (jobs->numel[i] < 0 || !jobs->param[i] || !jobs->grad[i] || !jobs->exp_avg[i] || !jobs->exp_avg_sq[i])
      return CGNN_EINVAL;
    blocks += (jobs->numel[i] + ADAM_PER_BLOCK - 1) / ADAM_PER_BLOCK;
  }
  if (blocks == 0) blocks = 1;                       // still advances the counter
  if (blocks > 0x7fffffff) return CGNN_EUNSUPPORTED;
  k_adam<<<(unsigned)blocks, ADAM_THR, 0, cgnn_stream(stream)>>>(*jobs, step, arrivals, advance ? 1 : 0, lr,
                                                                 beta1, beta2, (float)eps, (float)weight_decay);
  CGNN_CHECK_LAUNCH();
  return CGNN_OK;
}
