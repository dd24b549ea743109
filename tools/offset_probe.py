#!/usr/bin/env python3
"""Error of every gradient against the float64 oracle, HIP vs the fp32 oracle, for node features
with a large offset (layer-0 statistics / weight-gradient conditioning).  usage: offset_probe.py [offset scale]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import connectome_gnn_amd as C
from tests import parity as P

offset, scale = (float(sys.argv[1]), float(sys.argv[2])) if len(sys.argv) > 2 else (100.0, 1.0)
graphs = C.generate_dataset(12, 360, 14, seed=5)
b = C.collate_graphs(graphs)
b.node_features = b.node_features * scale + offset
for impl in ("auto", "layered"):
    torch.manual_seed(11)
    m = C.GCNConnectome(5, 64, dropout=0.0, impl=impl)
    sd0 = {k: v.clone() for k, v in m.state_dict().items()}
    lo, loss_o, g32, st32 = P.oracle_run("gcn", sd0, b)
    lo64, loss64, g64, st64 = P.oracle_run("gcn", sd0, b, dtype=torch.float64)
    m = m.to("cuda").train()
    bd = b.to("cuda")
    lg = m(bd)
    torch.nn.functional.cross_entropy(lg, bd.labels).backward()
    print(f"== impl {impl} ({m.impl_used}) offset {offset} scale {scale}")
    print(f"logits: hip {float((lg.cpu().double()-lo64).abs().max()):.2e} cpu32 {float((lo.double()-lo64).abs().max()):.2e}")
    for k, p in m.named_parameters():
        e_g = float((p.grad.cpu().double() - g64[k]).abs().max())
        e_c = float((g32[k].double() - g64[k]).abs().max())
        print(f"{k:28s} scale {float(g64[k].abs().max()):.2e}  hip {e_g:.2e}  cpu32 {e_c:.2e}  ratio {e_g / max(e_c, 1e-30):6.2f}")
    sd = m.state_dict()
    for k in st64:
        if "running" in k:
            e_g = float((sd[k].cpu().double() - st64[k]).abs().max())
            e_c = float((st32[k].double() - st64[k]).abs().max())
            print(f"{k:28s} scale {float(st64[k].abs().max()):.2e}  hip {e_g:.2e}  cpu32 {e_c:.2e}")
