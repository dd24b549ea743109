// common.h -- shared device helpers for the gfx950 kernels (wave64, CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/cgnn.h"

#define CGNN_WAVE 64

#define CGNN_CHECK_LAUNCH()                                   \
  do {                                                        \
    if (hipGetLastError() != hipSuccess) return CGNN_ELAUNCH; \
  } while (0)

// ABI 2: every scratch / partial-sum buffer the library writes comes with its byte count; a call whose
// buffer is shorter than what the path it is about to take writes returns CGNN_EINVAL before any launch.
#define CGNN_NEED_BYTES(ptr, have, need)                                   \
  do {                                                                     \
    if ((ptr) && (int64_t)(have) < (int64_t)(need)) return CGNN_EINVAL;    \
  } while (0)

static inline hipStream_t cgnn_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// Per-device lazily-initialised state (function attributes, CU counts) is indexed by the current
// device ordinal: one process may drive several GPUs.
#define CGNN_MAX_DEVICES 64
static inline int cgnn_device_ordinal() {
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= CGNN_MAX_DEVICES) return 0;
  return d;
}

static inline int64_t cgnn_align_up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }

// Wave-uniform value made provably uniform for the compiler (-> SGPR, scalar loads).
__device__ __forceinline__ int cgnn_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

__device__ __forceinline__ float cgnn_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double cgnn_wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Streamed data -- arrays a kernel reads or writes exactly once -- goes through non-temporal
// accesses: they do not allocate in the cache hierarchy the way default loads/stores do, and a
// streaming pass over 0.4 GB ran 24 % faster with them (k_pool_fwd 98 -> 74 us).  NOT for data
// that is re-read (tiles staged in LDS are fine: they are read from HBM once).
typedef float cgnn_f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t cgnn_u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t cgnn_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float4 ldnt4(const float* p) {
  const cgnn_f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const cgnn_f32x4*>(p));
  return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void stnt4(float* p, const float4& v) {
  __builtin_nontemporal_store(cgnn_f32x4{v.x, v.y, v.z, v.w}, reinterpret_cast<cgnn_f32x4*>(p));
}
__device__ __forceinline__ float ldnt(const float* p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ uint4 ldnt(const uint4* p) {
  const cgnn_u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const cgnn_u32x4*>(p));
  return make_uint4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ uint2 ldnt(const uint2* p) {
  const cgnn_u32x2 v = __builtin_nontemporal_load(reinterpret_cast<const cgnn_u32x2*>(p));
  return make_uint2(v[0], v[1]);
}
