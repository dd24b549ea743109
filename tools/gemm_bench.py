#!/usr/bin/env python3
"""Isolated timing of the fp32 MFMA projection kernels (cgnn_linear_*) at the batch shapes of the
BASELINE configs: achieved TFLOP/s against the 157.3 TFLOP/s fp32 matrix peak (MI355X_MICROARCH.md)
and the HBM-side rate (these GEMMs are tall-skinny: arithmetic intensity 2*K*N/(4*(K+N)) flop/B)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from connectome_gnn_amd import ops  # noqa: E402

PEAK_TF = 157.3
out = []
for name, m, k, n in (("cfg4 hidden layer", 4096 * 360, 64, 64), ("cfg3 SAGE layer", 512 * 360, 256, 128),
                      ("cfg5-like", 64 * 1000, 256, 256)):
    x = torch.randn(m, k, device="cuda")
    w = torch.randn(n, k, device="cuda") / k ** 0.5
    dy = torch.randn(m, n, device="cuda")
    dw = torch.empty_like(w)
    fns = {"fwd": lambda: ops.linear_fwd_raw(x, None, w, None, False),
           "bwd_input": lambda: ops.linear_bwd_input_raw(dy, w, 0, k),
           "bwd_weight": lambda: ops.linear_bwd_weight_raw(dy, x, dw, 0)}
    for tag, fn in fns.items():
        for _ in range(3):
            fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            fn()
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 10
        flops = 2.0 * m * k * n
        byts = 4.0 * (m * k + m * n + k * n)
        out.append({"shape": name, "M": m, "K": k, "N": n, "kernel": tag, "ms": round(ms, 4),
                    "TFLOPs": round(flops / ms / 1e9, 2), "frac_of_fp32_mfma_peak": round(flops / ms / 1e9 / PEAK_TF, 3),
                    "GBps": round(byts / ms / 1e6, 1)})
        print(out[-1])
json.dump(out, open(os.path.join(os.path.dirname(__file__), "..", "gpurun_out", "gemm_bench.json"), "w"), indent=1)
