// kernarg_probe.hip -- does the size of a kernel's by-value arguments cost launch time?
// (The kernel trace shows ~6 us of idle before AND after every k_gcn_bwd / k_agg_tiled launch and
// around no other kernel; those two have the largest argument blocks of the library.)
// build: hipcc -O2 --offload-arch=gfx950 tools/kernarg_probe.hip -o tools/_bin/kernarg_probe
#include <hip/hip_runtime.h>
#include <cstdio>
template <int N> struct Pay { char b[N]; };
template <int N> __global__ void k(Pay<N> p, int* out) {
  if (threadIdx.x == 0 && p.b[0] == 77 && p.b[N - 1] == 55) out[0] = 1;
}
__global__ void busy(int* out, int n) {
  int v = threadIdx.x;
  for (int i = 0; i < n; ++i) v = v * 1664525 + 1013904223;
  if (v == 12345) out[1] = v;
}
template <int N> void run(int* out) {
  Pay<N> p{};
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(a, 0);
    for (int i = 0; i < 300; ++i) {
      busy<<<256, 256>>>(out, 2000);          // ~ a few us of real work between the probed launches
      k<N><<<256, 64>>>(p, out);
    }
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
  }
  float ms = 0.f;
  hipEventElapsedTime(&ms, a, b);
  printf("explicit kernarg %4d B (+8 pointer): %7.2f us per (busy + probe) pair\n", N, ms * 1000.f / 300.f);
}
int main() {
  int* out;
  hipMalloc(&out, 64);
  run<16>(out); run<64>(out); run<128>(out); run<160>(out); run<184>(out); run<192>(out); run<200>(out);
  run<208>(out); run<216>(out); run<224>(out); run<240>(out); run<256>(out); run<272>(out); run<320>(out);
  run<512>(out); run<1024>(out);
  return 0;
}
