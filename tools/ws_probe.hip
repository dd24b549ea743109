// Diagnostic harness: times the weight-stationary kernels of gemm_ws.hip in isolation with
// -DWS_PROBE_* variants.  hipcc --offload-arch=gfx950 -O3 [-DWS_PROBE_x] tools/ws_probe.hip -o tools/_bin/ws_probe_x
#include "../connectome_gnn_amd/csrc/gemm_ws.hip"
#include <cstdio>
extern "C" int cgnn_fused_grid(void) { return 256; }
int main() {
  const int64_t M = 512 * 360; const int K = 256, N = 128;
  float *X, *W, *Y, *dX, *slab;
  hipMalloc(&X, M * K * 4); hipMalloc(&W, N * K * 4); hipMalloc(&Y, M * N * 4); hipMalloc(&dX, M * K * 4);
  hipMalloc(&slab, 256ll * N * K * 4);
  hipMemset(X, 0, M * K * 4); hipMemset(W, 0, N * K * 4); hipMemset(Y, 0, M * N * 4);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  auto run = [&](const char* name, auto fn) {
    for (int i = 0; i < 3; ++i) fn();
    hipEventRecord(a);
    for (int i = 0; i < 10; ++i) fn();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%s %.1f us  (%.1f TFLOP/s)\n", name, ms * 100, 2.0 * M * K * N / (ms / 10) / 1e9);
  };
  run("fwd", [&] { cgnn_ws_linear_fwd(X, K, K, nullptr, 0, 0, W, nullptr, 0, Y, N, M, N, nullptr, 0); });
  run("bwd_input", [&] { cgnn_ws_linear_bwd_input(Y, N, W, K, 0, dX, K, M, N, K, 0); });
  run("bwd_weight", [&] { cgnn_ws_linear_bwd_weight(Y, N, X, K, K, nullptr, 0, 0, slab, M, N, 0); });
  return 0;
}
