#!/usr/bin/env python3
"""Where a fresh-batch replayed step's time goes (cfg2 shape by default): fixed-batch replay, the
subject-cache replay without / with the per-step id copy, and the whole Trainer.train_epoch loop."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import connectome_gnn_amd as C
from connectome_gnn_amd.graphed import GraphedTrainStep, GraphedResidentStep
from connectome_gnn_amd.optim import Adam
from connectome_gnn_amd.resident import ResidentDataLoader, assemble_batch
from connectome_gnn_amd.structure_cache import ResidentBatch, SubjectStructureCache
from connectome_gnn_amd.synthetic import generate_packed
from connectome_gnn_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=84)
ap.add_argument("--k", type=int, default=8)
ap.add_argument("--batch", type=int, default=512)
ap.add_argument("--iters", type=int, default=200)
a = ap.parse_args()
dev = torch.device("cuda", 0)
ds = generate_packed(4 * a.batch, a.n, a.k, seed=42).to(dev)


def timeit(fn, iters=a.iters):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def fresh():
    torch.manual_seed(42)
    m = C.GCNConnectome(5, 64, 2, 3, 0.3).to(dev).train()
    return m, Adam(m.parameters(), lr=1e-3, weight_decay=1e-4)


loss_fn = ops.CrossEntropyLoss()
m, opt = fresh()
b = assemble_batch(ds, torch.randperm(a.batch))
b.structure()
g = GraphedTrainStep(m, opt, b, loss_fn)
print(f"fixed batch, per-batch build ({int(b.structure().fused_meta(384, 256).tile_ptr.numel()) - 1} tiles): {timeit(g):.4f} ms")
del g
m, opt = fresh()
cache = SubjectStructureCache(ds)
rb = ResidentBatch(cache, torch.randperm(4 * a.batch)[:a.batch])
g = GraphedResidentStep(m, opt, rb, loss_fn)
print(f"subject cache, replay only: {timeit(lambda: g()):.4f} ms")
chunks = [torch.randperm(4 * a.batch)[:a.batch] for _ in range(8)]
rbs = [ResidentBatch(cache, c) for c in chunks]
i = [0]
def withcopy():
    i[0] += 1
    g(rbs[i[0] % 8])
print(f"subject cache, replay + id copy: {timeit(withcopy):.4f} ms")
def withbatch():
    i[0] += 1
    g(ResidentBatch(cache, chunks[i[0] % 8]))
print(f"subject cache, new ResidentBatch + replay: {timeit(withbatch):.4f} ms")
del g
m, opt = fresh()
tr = C.Trainer(m, opt, device="cuda", graph=True)
ld = ResidentDataLoader(ds, batch_size=a.batch, shuffle=True, structure_cache=True)
ld.structure_cache = cache
tr.train_epoch(ld)
print(f"Trainer.train_epoch: {timeit(lambda: tr.train_epoch(ld), 20) / len(ld):.4f} ms per step")
