#!/usr/bin/env python3
"""BASELINE config 5's measurement: the scatter (aggregation) kernel alone on 1000-ROI graphs at
10 % density, hidden 256, fp16 storage / fp32 accumulate -- achieved GB/s of algorithmic traffic
(2*Nn*F*s + 8*Ee + 4*(Nn+1) = 1.828 MB/graph) against the 8 TB/s HBM roofline, next to
the fp32 gather kernel (cgnn_aggregate_f32) on the same graphs."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from connectome_gnn_amd import _lib, ops  # noqa: E402
from connectome_gnn_amd.resident import assemble_batch  # noqa: E402
from connectome_gnn_amd.synthetic import generate_packed  # noqa: E402

B, N, K, H = 64, 1000, 100, 256
ds = generate_packed(B, N, K, seed=42).to("cuda")
b = assemble_batch(ds, torch.arange(B))
s = b.structure()
grid = _lib.load().cgnn_fused_grid()
meta = s.fused_meta(1024, grid, 1.0)
norm = s.gcn_norm()
x32 = torch.randn(s.num_nodes, H, device="cuda")
x16 = x32.half()


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return a.elapsed_time(e) / reps


def alg_bytes(sz):
    return 2.0 * s.num_nodes * H * sz + 8.0 * s.num_edges + 4.0 * (s.num_nodes + 1)


out = {"workload": f"cfg5 scatter: {B} x {N}-ROI graphs, k={K} ({s.num_edges // B} edges/graph), hidden {H}",
       "hbm_peak_GBps": 8000.0, "results": []}
mden = ops.dense_adj_f16(s, norm.coef_dst, norm.selfc)
mpk = ops.dense_pack_f16(s, norm.coef_dst, norm.selfc)
out["operator_MB"] = {"dense": round(mden.numel() * 2 / 1e6, 1), "per_fragment": round(mpk.nbytes() / 1e6, 1)}
for name, fn, sz in (
        ("cgnn_dense_aggregate_c16 (fp16 storage, per-fragment operator on the fp16 matrix cores)",
         lambda: ops.dense_aggregate_c16_raw(s, mpk, x16), 2),
        ("cgnn_dense_aggregate_f16 (fp16 storage, dense M_g on the fp16 matrix cores)",
         lambda: ops.dense_aggregate_f16_raw(s, mden, x16), 2),
        ("cgnn_aggregate_tiled_f16 (fp16 storage, LDS tiles)",
         lambda: ops.aggregate_tiled_f16_raw(s, meta, 0, x16, norm.dis, norm.dis, None), 2),
        ("cgnn_aggregate_tiled_f16 transposed",
         lambda: ops.aggregate_tiled_f16_raw(s, meta, ops.AGG_TRANSPOSED, x16, norm.dis, norm.dis, None), 2),
        ("cgnn_aggregate_f32 (fp32, gather from L2/HBM)",
         lambda: ops.aggregate_raw(s.rowptr_dst, s.col_dst, norm.coef_dst, norm.selfc, None, None, x32), 4)):
    ms = timed(fn)
    gbs = alg_bytes(sz) / ms / 1e6
    out["results"].append({"kernel": name, "ms": round(ms, 4), "algorithmic_MB": round(alg_bytes(sz) / 1e6, 2),
                           "GBps": round(gbs, 1), "frac_of_hbm_peak": round(gbs / 8000.0, 4)})
    print(out["results"][-1])
os.makedirs(os.path.join(os.path.dirname(__file__), "..", "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(os.path.dirname(__file__), "..", "gpurun_out", "scatter_bench.json"), "w"), indent=1)
