#!/usr/bin/env python3
"""cfg5 dense fp16 aggregate alone (64 x 1000-ROI, 256 columns), with a 512 MiB write between
calls so that M_g comes from HBM as it does inside the training step.  CGNN_LIB selects the build."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from connectome_gnn_amd import _lib, ops  # noqa: E402
from connectome_gnn_amd.resident import assemble_batch  # noqa: E402
from connectome_gnn_amd.synthetic import generate_packed  # noqa: E402

ds = generate_packed(64, 1000, 100, seed=42).to("cuda")
b = assemble_batch(ds, torch.arange(64))
s = b.structure()
norm = s.gcn_norm()
x16 = torch.randn(s.num_nodes, 256, device="cuda").half()
mden = ops.dense_adj_f16(s, norm.coef_dst, norm.selfc)
pk = ops.dense_pack_f16(s, norm.coef_dst, norm.selfc)
junk = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
for name, fn in (("dense_agg ", lambda: ops.dense_aggregate_f16_raw(s, mden, x16)),
                 ("packed_agg", lambda: ops.dense_aggregate_c16_raw(s, pk, x16))):
    for cold in (False, True):
        ts = []
        for _ in range(12):
            if cold:
                junk.fill_(1)
            a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            fn()
            e.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(e) * 1e3)
        ts = sorted(ts[2:])
        print(f"  {name} {'cold' if cold else 'warm'}: median {ts[len(ts) // 2]:7.1f} us")
