"""Does the row stride of the streamed operand matter for the weight-stationary kernels?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from connectome_gnn_amd import ops
m, k, n = 512 * 360, 256, 128
w = torch.randn(n, k, device="cuda") / 16
def t(fn):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / 10 * 1e3
for ld in (256, 260, 272, 288, 320):
    xb = torch.randn(m, ld, device="cuda"); x = xb[:, :k]
    print("fwd single panel ld", ld, round(t(lambda: ops.linear_fwd_raw(x, None, w, None, False)), 1), "us")
for ld in (128, 132, 144, 160):
    x1 = torch.randn(m, ld, device="cuda")[:, :128]; x2 = torch.randn(m, ld, device="cuda")[:, :128]
    print("fwd two panels ld", ld, round(t(lambda: ops.linear_fwd_raw(x1, x2, w, None, True)), 1), "us")
