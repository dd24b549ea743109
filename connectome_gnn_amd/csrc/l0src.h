// l0src.h -- rebuild rows of layer 0's output from the narrow aggregate (include/cgnn.h, cgnn_l0src).
//
// Y0 = P0 W0^T + b0 with P0 = A_hat X0 [Nn, 8] is never written to HBM by the fused path: every
// consumer rebuilds the columns it needs: bias + fma chain, k ascending -- l0_rebuild4 on the
// vector ALUs (k_l0_bwd) or l0_mfma on the fp32 matrix pipe (the tile kernels' forward input and
// backward X_prev, which therefore agree bit for bit: same instruction, same operands).
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

// The same rows on the fp32 matrix pipe (v_mfma_f32_16x16x4_f32: D = C + A B, an fmaf chain over
// k like the expression above): a 16 x 16 block of Y0 is two MFMAs (k = 0..3, 4..7) with the bias
// as C.  L0W holds this lane's W0 fragments for four blocks:
//   l0w_cols(c) : block c = columns 16c + (lane&15)   (W0 as the M operand: D[col][row])
//   l0w_quad(t) : block t = columns 4*(lane&15) + t   (W0 as the N operand: D[row][col])
// and the P0 operand of lane (i = lane&15, kq = lane>>4) is P0[row i][kq], P0[row i][4 + kq].
typedef float l0_f32x4 __attribute__((ext_vector_type(4)));
struct L0W { float a[4], b[4]; };

__device__ __forceinline__ L0W l0w_cols(const cgnn_l0src& l0, int lane) {
  const int i = lane & 15, kq = lane >> 4;
  L0W w;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    w.a[c] = kq < l0.F0 ? l0.W0[(16 * c + i) * l0.F0 + kq] : 0.f;
    w.b[c] = 4 + kq < l0.F0 ? l0.W0[(16 * c + i) * l0.F0 + 4 + kq] : 0.f;
  }
  return w;
}
__device__ __forceinline__ L0W l0w_quad(const cgnn_l0src& l0, int lane) {
  const int i = lane & 15, kq = lane >> 4;
  L0W w;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    w.a[t] = kq < l0.F0 ? l0.W0[(4 * i + t) * l0.F0 + kq] : 0.f;
    w.b[t] = 4 + kq < l0.F0 ? l0.W0[(4 * i + t) * l0.F0 + 4 + kq] : 0.f;
  }
  return w;
}
// D[m][n] = bias + sum_k M[m][k] N[k][n]; (ma, mb) / (na, nb): this lane's k = kq / 4 + kq values
__device__ __forceinline__ l0_f32x4 l0_mfma(float ma, float mb, float na, float nb, l0_f32x4 c) {
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(ma, na, c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_16x16x4f32(mb, nb, c, 0, 0, 0);
}

}  // namespace
