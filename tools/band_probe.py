#!/usr/bin/env python3
"""The large-graph fp32 aggregation on 64 x 1000-ROI graphs, F = 256, back to back: gather over all edges, gather
over the edges outside the dense fragments, the dense-fragment kernel alone, and the two-launch form."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import connectome_gnn_amd as C
from connectome_gnn_amd import ops
from connectome_gnn_amd.resident import assemble_batch
from connectome_gnn_amd.synthetic import generate_packed
dev = torch.device("cuda:0")
from connectome_gnn_amd import _lib
def _band_only(s, op, x, y):
    lib = _lib.load()
    _lib.check(lib.cgnn_band_aggregate_f32(_lib.ptr(op.bfrag), _lib.ptr(op.bstep), _lib.ptr(op.boff), op.pitch, _lib.ptr(s.gptr),
               s.num_graphs, _lib.ptr(x), x.stride(0), x.shape[1], None, None, 0, _lib.ptr(y), y.stride(0), _lib.stream_ptr()), "band")

ds = generate_packed(64, 1000, 100, seed=1)
b = assemble_batch(ds, torch.arange(64)).to(dev)
s = b.structure()
nrm = s.gcn_norm()
F = 256
x = torch.randn(s.num_nodes, F, device=dev)
for name, rp, col, coef in (("fwd", s.rowptr_dst, s.col_dst, nrm.coef_dst), ("bwd", s.rowptr_src, s.col_src, nrm.coef_src)):
    op = ops.band_operator_f32(s, rp, col, coef)
    print(name, "items", op.num_items, "covered %.3f" % op.covered, "rem edges", op.coef.numel(), "of", coef.numel(), flush=True)
    y = torch.empty_like(x)
    def t(fn, n=30):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    print("  gather all  %.1f us" % t(lambda: ops.aggregate_raw(rp, col, coef, nrm.selfc, None, None, x, out=y)))
    print("  gather rem  %.1f us" % t(lambda: ops.aggregate_raw(op.rowptr, op.col, op.coef, nrm.selfc, None, None, x, out=y)))
    print("  band        %.1f us" % t(lambda: _band_only(s, op, x, y)))
    print("  both        %.1f us" % t(lambda: ops.aggregate_raw(rp, col, coef, nrm.selfc, None, None, x, out=y, band=(s, op))), flush=True)
