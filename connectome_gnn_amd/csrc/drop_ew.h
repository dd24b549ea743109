// drop_ew.h -- dropout keep bits of the elementwise / layered kernels (one byte per 4-column chunk,
// keyed counter hash over the chunk index); shared by elementwise.hip and aggregate_tiled.hip so
// that a fused consumer draws exactly the bits the stand-alone apply pass would.
#pragma once
#include "common.h"

namespace {

struct DropCfg {
  uint32_t thr16;
  float scale;
  uint32_t key0, key1;
  const uint32_t* dev_key;   // optional device word XOR-ed into key1 (fresh masks per graph replay)
};

__device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}

__device__ __forceinline__ uint32_t drop_bits(const DropCfg& d, uint32_t chunk_index) {
  // one keyed counter hash + one chained round = 64 random bits (as in fused_gcn.hip)
  const uint32_t h0 = mix32((chunk_index ^ d.key0) + d.key1);
  const uint32_t h1 = mix32(h0 + 0x9E3779B9u);
  uint32_t b = 0;
  b |= ((h0 & 0xFFFFu) >= d.thr16) ? 1u : 0u;
  b |= ((h0 >> 16) >= d.thr16) ? 2u : 0u;
  b |= ((h1 & 0xFFFFu) >= d.thr16) ? 4u : 0u;
  b |= ((h1 >> 16) >= d.thr16) ? 8u : 0u;
  return b;
}

// One element of BatchNorm's backward, dY = a * ((g*f - c1) - xhat*c2) with xhat = (y - mean) * invstd,
// in ONE fixed sequence of separately rounded operations: k_bn_act_apply<true> (elementwise.hip) and
// the aggregate that forms dY while staging (dense_aggregate_c16.hip) must produce the same bits, and
// neither `#pragma clang fp contract(off)` nor expression shape survives inlining into kernels that
// are otherwise compiled with contraction on -- so the operations are spelled as instructions.
__device__ __forceinline__ float ew_mul(float a, float b) {
  float r;
  asm("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float ew_sub(float a, float b) {
  float r;
  asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float bn_bwd_dy(float a, float g, float f, float c1, float y, float mean,
                                           float invstd, float c2) {
  const float xhat = ew_mul(ew_sub(y, mean), invstd);
  return ew_mul(a, ew_sub(ew_sub(ew_mul(g, f), c1), ew_mul(xhat, c2)));
}

inline DropCfg make_drop(float p, uint64_t seed, int* use_drop) {
  DropCfg d;
  *use_drop = (p > 0.f) ? 1 : 0;
  double thr = (double)p * 65536.0 + 0.5;
  if (thr > 65535.0) thr = 65535.0;
  d.thr16 = (uint32_t)thr;
  // the reference's scale, 1/(1-p) (aten::native_dropout), not 1/(realised keep rate): with
  // replayed keep bits the arithmetic then matches the oracle to rounding
  d.scale = p > 0.f ? (float)(1.0 / (1.0 - (double)p)) : 1.0f;
  d.key0 = (uint32_t)(seed & 0xFFFFFFFFu) * 0x9E3779B9u + 0x85EBCA6Bu;
  d.key1 = (uint32_t)(seed >> 32) ^ 0xC2B2AE35u;
  d.dev_key = nullptr;
  return d;
}

}  // namespace
