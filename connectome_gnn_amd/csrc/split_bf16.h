// split_bf16.h -- exact fp32 products on the bf16 matrix pipe of gfx950.
//
// The fp32 MFMA pipe runs at 1/16 of the bf16 rate.  A float splits EXACTLY into three bf16 pieces
// by truncation (24 significant bits = 8 + 8 + 8):  x = h + m + l.  A product x*w is then the six
// partial products of total order <= 2
//     h*h' + (h*m' + m*h') + (h*l' + m*m' + l*h')
// each exact in fp32 (8 x 8 significant bits) and accumulated in fp32 by the bf16 MFMA; the three
// dropped terms are <= 2^-24 of the product, i.e. at the rounding level of an fp32 multiply-add.
#pragma once
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ uint32_t pack_hi16(float lo, float hi) {   // (bf16(lo), bf16(hi)), truncating
  return __builtin_amdgcn_perm(__float_as_uint(hi), __float_as_uint(lo), 0x07060302u);
}
__device__ __forceinline__ float trunc_bf16(float x) { return __uint_as_float(__float_as_uint(x) & 0xFFFF0000u); }

struct Split8 { uint4 h, m, l; };     // 8 values (two float4) as three bf16x8 fragments

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
// (bf16(lo), bf16(hi)) rounded to nearest even: one v_cvt_pk_bf16_f32
__device__ __forceinline__ uint32_t pack_rn16(float lo, float hi) {
  const f32x2 v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}

#ifndef CGNN_SPLIT_RN
__device__ __forceinline__ Split8 split8(const float4& a, const float4& b) {
  const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  uint32_t ph[4], pm[4], pl[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const float x0 = x[2 * p], x1 = x[2 * p + 1];
    const float r0 = x0 - trunc_bf16(x0), r1 = x1 - trunc_bf16(x1);
    const float l0 = r0 - trunc_bf16(r0), l1 = r1 - trunc_bf16(r1);
    ph[p] = pack_hi16(x0, x1);
    pm[p] = pack_hi16(r0, r1);
    pl[p] = pack_hi16(l0, l1);
  }
#else
// DIAGNOSTIC VARIANT (make variant TAG=rn DEFS=-DCGNN_SPLIT_RN; never the product).  Round-to-nearest pieces:
// h = rn(x), m = rn(x - h), l = x - h - m.  Still exact (x - h has <= 16 significant bits, x - h - m <= 8), same
// instruction count as the truncating form (v_cvt_pk_bf16_f32 packs and rounds two values), and m, l carry
// either sign, so the three dropped terms m*l', l*m', l*l' of a product are two-sided.  Measured in round 4 on
// one box, A/B/A/B: GraphSAGE h128 step 1.335 -> 1.394 ms (+4.4 %: the cvt -> shift -> sub -> cvt chain is two
// instructions deeper than and -> sub -> and), cfg5-in-fp32 +3.1 %, headline +0.5 %; and ReLU decisions within
// rounding of zero do not become rarer, they move (one parity case gained a tie, another lost its own): the
// ties come from fp32 ACCUMULATION order (~1e-7 of the sum of magnitudes), the dropped terms are 2^-24 of
// single products.  Not kept.
__device__ __forceinline__ Split8 split8(const float4& a, const float4& b) {
  const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  uint32_t ph[4], pm[4], pl[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const float x0 = x[2 * p], x1 = x[2 * p + 1];
    ph[p] = pack_rn16(x0, x1);
    const float r0 = x0 - __uint_as_float(ph[p] << 16), r1 = x1 - __uint_as_float(ph[p] & 0xFFFF0000u);
    pm[p] = pack_rn16(r0, r1);
    const float l0 = r0 - __uint_as_float(pm[p] << 16), l1 = r1 - __uint_as_float(pm[p] & 0xFFFF0000u);
    pl[p] = pack_hi16(l0, l1);         // (exactly representable: truncation == rounding)
  }
#endif
  Split8 s;
  s.h = make_uint4(ph[0], ph[1], ph[2], ph[3]);
  s.m = make_uint4(pm[0], pm[1], pm[2], pm[3]);
  s.l = make_uint4(pl[0], pl[1], pl[2], pl[3]);
  return s;
}

}  // namespace
