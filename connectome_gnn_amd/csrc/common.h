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
