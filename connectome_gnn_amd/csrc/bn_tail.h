// bn_tail.h -- batch-wide BatchNorm sums WITHOUT a launch of their own (round 4).
//
// Every tile kernel of the per-tile GCN path ends with per-workgroup fp64 partials of 128 column sums
// (sum y | sum y^2 in the forward pass, sum dz | sum dz*xhat in the backward pass).  Rounds 1-3 wrote them
// to a [workgroups][128] slab and a separate one-block-per-channel kernel folded the slab and turned the
// sums into the layer's coefficient block: five launches of 4-7 us per training step that do nothing but
// wait for one load round trip.
//
// Here the partials go into an accumulator with 64-bit ATOMIC adds and the workgroup that arrives last (an
// arrival counter) finalises the layer in the producer's own tail:
//
//   * the sums are added as FIXED-POINT integers (integer part and fractional part x 2^52 in two signed
//     64-bit words): integer addition is associative, so the result does not depend on the order in which
//     the workgroups arrive -- bit-identical reruns, which fp64 atomic adds would not give (tools/tail_stress.py:
//     600 training steps twice, every gradient and running statistic equal, at four batch sizes).  Against the
//     slab protocol's fp64 tree fold of 256 rounded partials the exact sum differs in the last bits: the same
//     coefficient block on small batches, gradients within 2e-7 of each other at 512 .. 4096 graphs;
//   * no fence: a returning atomic has been performed at the device's coherence point when its value comes
//     back, so "wait for my returns, then bump the counter" orders a workgroup's sums before its arrival
//     without the release fence whose L2 write-back made the round-2 "last workgroup" tail slower than the
//     separate kernels (it also read a 262 KB slab; the accumulator is 16 KB);
//   * eight sub-accumulators (workgroup % 8): 256 workgroups adding to ONE set of 256 addresses serialise at
//     the memory-side atomic unit -- the first version (one set, a 128-bit value with a carry between its two
//     atomics = two dependent round trips) made every producer ~8 us longer and LOST to the separate kernels
//     (cfg2 replay 0.183 vs 0.167 ms, 512-graph shard 0.342 vs 0.331, headline 1.864 vs 1.844);
//   * the last workgroup reads the accumulator with agent-scope atomic loads, writes the coefficient block
//     (+ running statistics, num_batches_tracked, the step's fresh dropout words) and leaves the accumulator
//     ZERO again, so one allocation serves every step and every HIP-graph replay;
//   * non-finite partials cannot be represented: they raise a flag word and the tail writes NaN sums, as the
//     slab path would have.
//
// Measured (A/B/A/B on one box, CGNN_DIAG_NO_BN_TAILS=1 = the slab protocol): headline 1.842 vs 1.856 ms
// (-0.8 %), 512-graph shard 0.328 vs 0.329, cfg2 replay 0.170 vs 0.167: three serialised round trips at the end
// of a producer (sums, arrival, the last workgroup's reads) cost what the launch they replace cost, so under
// HIP-graph replay it is a wash at small batches; what it buys is five launches fewer for eager callers and
// the slabs' traffic at the headline size.
#pragma once
#include "common.h"

namespace {

// Eight sub-accumulators (workgroup % 8, i.e. one per XCD under round-robin dispatch) cut the contention on
// an address eightfold; a column's sum is two independent signed 64-bit words -- integer part and fractional
// part scaled by 2^52 (exact below the resolution 2^-52, far under BatchNorm's eps) -- so the two atomics of a
// column have no carry between them and cost ONE round trip.
constexpr int BN_SUB = 8;
struct BnAcc {
  long long hi[BN_SUB][128];    // sum of floor(x)
  long long lo[BN_SUB][128];    // sum of (x - floor(x)) * 2^52
  unsigned int arrivals;
  unsigned int nonfinite;
  unsigned int pad[14];
};
static_assert(sizeof(BnAcc) == CGNN_BN_ACC_BYTES, "cgnn.h: CGNN_BN_ACC_BYTES");
static_assert(sizeof(cgnn_bn_tail) == 120, "cgnn.h: struct cgnn_bn_tail (ctypes mirror: _lib.CgnnBnTail)");

#define CGNN_AGENT __HIP_MEMORY_SCOPE_AGENT

// column `c`'s partial of this workgroup -> accumulator.  Returns a value that depends on both atomics'
// return values (consume it, e.g. store it to LDS, before bnacc_arrive: that is the wait).
__device__ __forceinline__ unsigned long long bnacc_add(BnAcc* acc, int c, double x) {
  if (!(fabs(x) < 4.0e18)) {                       // inf / nan / beyond 2^62: flagged, not added
    return (unsigned long long)__hip_atomic_fetch_or(&acc->nonfinite, 1u, __ATOMIC_RELAXED, CGNN_AGENT);
  }
  const int sub = blockIdx.x & (BN_SUB - 1);
  const double fl = floor(x);
  const long long hi = (long long)fl;
  const long long lo = (long long)((x - fl) * 4503599627370496.0);                // [0, 2^52)
  const long long a = __hip_atomic_fetch_add(&acc->hi[sub][c], hi, __ATOMIC_RELAXED, CGNN_AGENT);
  const long long b = __hip_atomic_fetch_add(&acc->lo[sub][c], lo, __ATOMIC_RELAXED, CGNN_AGENT);
  return (unsigned long long)(a ^ b);
}

// Call from ALL threads of the workgroup after every bnacc_add of the workgroup has returned (its return
// value consumed) and a __syncthreads().  True in every thread of the workgroup that arrived last.
__device__ __forceinline__ bool bnacc_arrive(BnAcc* acc, int* lds_flag) {
  if (threadIdx.x == 0) {
    const unsigned int old = __hip_atomic_fetch_add(&acc->arrivals, 1u, __ATOMIC_RELAXED, CGNN_AGENT);
    *lds_flag = (old == gridDim.x - 1) ? 1 : 0;
  }
  __syncthreads();
  return *lds_flag != 0;
}

// the finished sum of column c; the words are left zero (last workgroup only)
__device__ __forceinline__ double bnacc_take(BnAcc* acc, int c) {
  long long hi = 0, lo = 0;
#pragma unroll
  for (int s = 0; s < BN_SUB; ++s) {
    hi += __hip_atomic_load(&acc->hi[s][c], __ATOMIC_RELAXED, CGNN_AGENT);
    lo += __hip_atomic_load(&acc->lo[s][c], __ATOMIC_RELAXED, CGNN_AGENT);
  }
  const unsigned int bad = __hip_atomic_load(&acc->nonfinite, __ATOMIC_RELAXED, CGNN_AGENT);
#pragma unroll
  for (int s = 0; s < BN_SUB; ++s) {
    __hip_atomic_store(&acc->hi[s][c], 0ll, __ATOMIC_RELAXED, CGNN_AGENT);
    __hip_atomic_store(&acc->lo[s][c], 0ll, __ATOMIC_RELAXED, CGNN_AGENT);
  }
  if (bad) return __builtin_nan("");
  return (double)hi + (double)lo * 2.220446049250313e-16;          // 2^-52
}

// after every bnacc_take of the tail (one thread): counter and flag back to zero
__device__ __forceinline__ void bnacc_reset(BnAcc* acc) {
  __hip_atomic_store(&acc->arrivals, 0u, __ATOMIC_RELAXED, CGNN_AGENT);
  __hip_atomic_store(&acc->nonfinite, 0u, __ATOMIC_RELAXED, CGNN_AGENT);
}

__device__ __forceinline__ uint32_t bn_tail_mix32(uint32_t x) {      // (= mix32 of the dropout hash)
  x ^= x >> 16;
  x *= 0x7FEB352Du;
  x ^= x >> 15;
  x *= 0x846CA68Bu;
  x ^= x >> 16;
  return x;
}

// The finalisation itself, thread c < 64 of the last workgroup (the arithmetic of k_bn_fwd_stats /
// k_bn_bwd_stats, fused_gcn.hip).  mean_off: the constant the statistics were taken without (centred layer 0).
__device__ __forceinline__ void bn_tail_finalize(const cgnn_bn_tail& t, BnAcc* acc, int c, float mean_off) {
  const double S1 = bnacc_take(acc, c), S2 = bnacc_take(acc, 64 + c);
  if (t.mode == 0) {
    const double m = S1 / t.count;
    double v = S2 / t.count - m * m;
    if (v < 0.0) v = 0.0;
    const float mean = (float)m, var = (float)v;
    const double unbiased = t.count > 1.0 ? v * t.count / (t.count - 1.0) : v;
    const float mean_y = (float)(m + (double)mean_off);
    t.running_mean[c] = (1.0f - t.momentum) * t.running_mean[c] + t.momentum * (mean_off != 0.f ? mean_y : mean);
    t.running_var[c] = (1.0f - t.momentum) * t.running_var[c] + t.momentum * (float)unbiased;
    const float invstd = 1.0f / sqrtf(var + t.eps);
    const float a = t.gamma[c] * invstd;
    t.bn_out[c] = a;
    t.bn_out[64 + c] = t.beta[c] - mean * a;
    t.bn_out[128 + c] = mean;
    t.bn_out[192 + c] = invstd;
    if (c == 0 && t.num_batches_tracked) *t.num_batches_tracked += 1;
    if (t.rng_state && c < t.rng_n)
      t.rng_state[c] = bn_tail_mix32(t.rng_state[c] + 0x9E3779B9u * (uint32_t)(c + 1));
  } else {
    t.dbeta[c] = (float)S1;
    t.dgamma[c] = (float)S2;
    t.bwc[c] = t.zero_coef ? 0.f : (float)(S1 / t.count);
    t.bwc[64 + c] = t.zero_coef ? 0.f : (float)(S2 / t.count);
  }
}

// The whole tail for a kernel whose threads t < 128 hold the workgroup's partial of column t in wg_sums[t]
// (LDS, 128 doubles, written and synchronised by the caller); `scratch` >= 129 ints of LDS the caller no
// longer needs.  Call from ALL threads.  mean_off_of(c): the centred layer 0's constant (0 elsewhere).
template <typename MeanOff>
__device__ __forceinline__ void bn_tail_run(const cgnn_bn_tail& t, const double* wg_sums, int* scratch,
                                            MeanOff mean_off_of) {
  BnAcc* acc = static_cast<BnAcc*>(t.acc);
  if (threadIdx.x < 128) {
    const unsigned long long r = bnacc_add(acc, threadIdx.x, wg_sums[threadIdx.x]);
    scratch[1 + threadIdx.x] = (int)(r & 1u);      // consuming the returns = waiting for the atomics
  }
  __syncthreads();
  if (!bnacc_arrive(acc, scratch)) return;
  if (threadIdx.x < 64) bn_tail_finalize(t, acc, threadIdx.x, mean_off_of(threadIdx.x));
  __syncthreads();
  if (threadIdx.x == 0) bnacc_reset(acc);
}

}  // namespace
