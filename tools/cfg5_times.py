#!/usr/bin/env python3
"""Per-kernel times of the fp16-storage projections (gemm_h16.hip) at config 5's layer shape,
HIP events around back-to-back launches.  usage: python tools/cfg5_times.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from connectome_gnn_amd import ops

def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

M = 64000
x = torch.randn(M, 256, device="cuda").half(); dy = torch.randn(M, 256, device="cuda").half()
w = torch.randn(256, 256, device="cuda") / 16; b = torch.randn(256, device="cuda")
p0 = torch.randn(M, 64, device="cuda").half(); w0 = torch.randn(256, 5, device="cuda")
wh = w.half()
gb = 2 * M * 256 * 2 / 1e9
for name, fn, byts in (
    ("fwd  K256 N256 (ours)", lambda: ops.linear_fwd_f16_raw(x, w, b), gb),
    ("fwd  K256 N256 (hipBLASLt)", lambda: torch.matmul(x, wh.t()), gb),
    ("bwdi K256 N256 (ours)", lambda: ops.linear_bwd_input_f16_raw(dy, w), gb),
    ("bwdi K256 N256 (hipBLASLt)", lambda: torch.matmul(dy, wh), gb),
    ("bwdw K256 N256 (ours)", lambda: ops.linear_bwd_weight_f16_raw(dy, x), gb),
    ("fwd  K64(5) N256 (ours)", lambda: ops.linear_fwd_f16_raw(p0, w0, b), M * (64 + 256) * 2 / 1e9),
    ("bwdw K64(5) N256 (ours)", lambda: ops.linear_bwd_weight_f16_raw(dy, p0, 5), M * (64 + 256) * 2 / 1e9),
):
    us = timeit(fn)
    print(f"{name:30s} {us:8.1f} us  {byts / us * 1e6 / 1e3:6.2f} TB/s", flush=True)
