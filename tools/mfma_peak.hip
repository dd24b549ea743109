// Sustained fp32 matrix-core rate of this GPU: a register-only v_mfma_f32_32x32x2_f32 loop
// (4 independent accumulators per wave, WAVES waves per CU).  Prints TFLOP/s and the shader
// clock implied by clock64()/wall_clock64().  Diagnostic only (not part of the library):
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o gpurun_out/mfma_peak && gpurun_out/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void __launch_bounds__(512) k(int iters, float* out, unsigned long long* clk) {
  f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
  float x = threadIdx.x * 1e-3f, y = blockIdx.x * 1e-3f;
  const unsigned long long c0 = clock64(), w0 = wall_clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, a3, 0, 0, 0);
    }
  }
  const unsigned long long c1 = clock64(), w1 = wall_clock64();
  float s = 0;
  for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}
int main() {
  int cus = 0; hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  int wclk = 0; hipDeviceGetAttribute(&wclk, hipDeviceAttributeWallClockRate, 0);
  float* out; unsigned long long* clk;
  hipMalloc(&out, cus * 512 * 4); hipMalloc(&clk, 16);
  for (int waves = 4; waves <= 8; waves += 4) {
    const int iters = 20000;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<<<cus, waves * 64>>>(100, out, clk); hipDeviceSynchronize();
    hipEventRecord(a); k<<<cus, waves * 64>>>(iters, out, clk); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double flops = (double)cus * waves * iters * 64.0 * 4096.0;
    printf("cus %d waves/CU %d: %.3f ms  %.1f TFLOP/s  shader clock %.0f MHz (wall clock rate %d kHz)\n", cus,
           waves, ms, flops / ms / 1e9, (double)h[0] / ((double)h[1] / wclk) / 1e3, wclk);
  }
  return 0;
}
