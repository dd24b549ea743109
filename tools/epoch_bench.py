#!/usr/bin/env python3
"""End-to-end training throughput on FRESH batches: Trainer.train_epoch over a
ResidentDataLoader that re-shuffles every epoch, so on-device assembly, the CSR / blocked-ELL
builds and the step all count.  usage: tools/epoch_bench.py [num_subjects=32768] [batch=4096]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import connectome_gnn_amd as C  # noqa: E402
from connectome_gnn_amd.resident import ResidentDataLoader  # noqa: E402
from connectome_gnn_amd.synthetic import generate_packed  # noqa: E402

n_subj = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
bsz = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
ds = generate_packed(n_subj, 360, 14, seed=1).to("cuda")
epochs = 5
for prefetch in (False, True):
    torch.manual_seed(0)
    m = C.GCNConnectome(5, 64)
    tr = C.Trainer(m, torch.optim.Adam(m.parameters(), lr=1e-3, weight_decay=1e-4, fused=True), device="cuda")
    ld = ResidentDataLoader(ds, batch_size=bsz, shuffle=True, prefetch=prefetch, prepare=tr.model.prepare_batch)
    first = tr.train_epoch(ld)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(epochs):
        last = tr.train_epoch(ld)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    steps = epochs * len(ld)
    print(f"prefetch={prefetch}: {epochs * n_subj / dt:,.0f} graphs/s, {dt / steps * 1e3:.2f} ms per {bsz}-graph step "
          f"(loss {first:.4f} -> {last:.4f})")
