#!/usr/bin/env python3
"""Timing of the tiled LDS aggregate at the cfg3 GraphSAGE shape (512 x 360-ROI, 128 columns):
forward (post-divide) and backward (transposed, pre-divide, + Yadd).  CGNN_LIB selects the build."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from connectome_gnn_amd import _lib, ops  # noqa: E402
from connectome_gnn_amd.resident import assemble_batch  # noqa: E402
from connectome_gnn_amd.synthetic import generate_packed  # noqa: E402

lib = _lib.load()
ds = generate_packed(512, 360, 14, seed=1).to("cuda")
b = assemble_batch(ds, torch.arange(512, device="cuda"))
s = b.structure()
ell = s.fused_meta(384, int(lib.cgnn_fused_grid()), 0.0)
norm = s.sage_norm(backward_coef=False)
h = int(os.environ.get("H", 128))
x = torch.randn(s.num_nodes, h, device="cuda")
dcat = torch.randn(s.num_nodes, 2 * h, device="cuda")
fns = {"fwd  A X / den": lambda: ops.aggregate_tiled_raw(s, ell, ops.AGG_POST_DIV, x, None, norm.den, None),
       "bwd  X1 + A^T (dA / den)": lambda: ops.aggregate_tiled_raw(s, ell, ops.AGG_TRANSPOSED | ops.AGG_PRE_DIV,
                                                                   dcat[:, h:], norm.den, None, None, yadd=dcat[:, :h])}
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
