// l0src.h -- rebuild rows of layer 0's output from the narrow aggregate (include/cgnn.h, cgnn_l0src).
//
// Y0 = P0 W0^T + b0 with P0 = A_hat X0 [Nn, 8] is never written to HBM by the fused path: every
// consumer calls l0_rebuild4 for the 4 columns it needs.  ONE expression (b + fma chain, k
// ascending) everywhere, so the forward statistics, the next layer's input and the backward's
// xhat are computed from identical bits.
#pragma once
#include "common.h"

namespace {

constexpr int L0_FP = 8;                     // padded feature count of P0

// W0^T in LDS: wl[k][col] = W0[col][k] (zero for k >= F0), b0 behind it: wl[8*64 .. 8*64+63]
constexpr int L0_LDS_FLOATS = L0_FP * 64 + 64;

__device__ __forceinline__ void l0_stage(float* wl, const cgnn_l0src& l0, int nthreads) {
  for (int i = threadIdx.x; i < L0_FP * 64; i += nthreads) {
    const int k = i >> 6, c = i & 63;
    wl[i] = k < l0.F0 ? l0.W0[c * l0.F0 + k] : 0.f;
  }
  for (int i = threadIdx.x; i < 64; i += nthreads) wl[L0_FP * 64 + i] = l0.b0[i];
}

// columns col .. col+3 (col % 4 == 0) of the row whose narrow aggregate is (p0, p1)
__device__ __forceinline__ float4 l0_rebuild4(const float4& p0, const float4& p1, const float* wl,
                                              int col, int F0) {
  const float pv[L0_FP] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w};
  float4 y = *reinterpret_cast<const float4*>(wl + L0_FP * 64 + col);
#pragma unroll
  for (int k = 0; k < L0_FP; ++k) {
    if (k < F0) {                            // wave-uniform: columns >= F0 are zero
      const float4 wk = *reinterpret_cast<const float4*>(wl + k * 64 + col);
      y.x = fmaf(pv[k], wk.x, y.x); y.y = fmaf(pv[k], wk.y, y.y);
      y.z = fmaf(pv[k], wk.z, y.z); y.w = fmaf(pv[k], wk.w, y.w);
    }
  }
  return y;
}

}  // namespace
