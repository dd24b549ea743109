#!/usr/bin/env python3
"""Timing of the weight-stationary projection kernels at the cfg3 GraphSAGE layer shape
(M = 512 x 360, [X | A] 128 + 128 -> 128) through the C ABI: forward with statistics, backward
input, backward weight.  CGNN_LIB selects the library build (tools/ws_ab.sh)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from connectome_gnn_amd import _lib, ops, sage_path  # noqa: E402

lib = _lib.load()
m, h = 512 * 360, 128
x1, x2 = torch.randn(m, h, device="cuda"), torch.randn(m, h, device="cuda")
w = torch.randn(h, 2 * h, device="cuda") / 16
b = torch.randn(h, device="cuda")
dy = torch.randn(m, h, device="cuda")
dw = torch.empty_like(w)
if os.environ.get("ZERO"):      # all-zero operands: the clock the chip holds without MFMA data toggling
    for t in (x1, x2, w, b, dy):
        t.zero_()
grid = int(lib.cgnn_fused_grid())
ws = torch.empty(max(int(lib.cgnn_linear_bwd_weight2_workspace_bytes(m, h, h, h)), 16), dtype=torch.uint8, device="cuda")


def bwd_w():
    _lib.check(lib.cgnn_linear_bwd_weight2_f32(_lib.ptr(dy), h, _lib.ptr(x1), h, h, _lib.ptr(x2), h, h, _lib.ptr(dw),
                                               2 * h, m, h, _lib.ptr(ws), _lib.nbytes(ws), _lib.stream_ptr()), "bw")


fns = {"fwd+stats K=256 N=128": lambda: sage_path._linear_fwd_stats(lib, x1, x2, w, b, grid),
       "bwd_input N=128 K=256": lambda: ops.linear_bwd_input_raw(dy, w, 0, 2 * h),
       "bwd_weight K=256 N=128": bwd_w}
for tag, fn in fns.items():
    for _ in range(3):
        fn()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        fn()
    e.record()
    torch.cuda.synchronize()
    print(f"  {tag:26s} {a.elapsed_time(e) / 20 * 1e3:8.1f} us")
