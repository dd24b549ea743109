#!/usr/bin/env python3
"""bench.py -- training graphs/sec of the batched message-passing path on MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one full training step of the reference's loop (train.py:46-51): zero_grad,
forward (normalisation + 3 layers + pool + head), CrossEntropy, backward, gradient all-reduce
(N > 1), Adam(lr 1e-3, wd 1e-4) -- on a batch that is already resident in HBM with its CSR
structure built (the reference's collate is outside its step as well).
Default workload = BASELINE.json's headline: 3-layer GCN, hidden 64, a global batch of 4096
graphs x 360 ROI (Watts-Strogatz k=14, beta 0.15), fp32, dropout 0.3.

Multi-GPU (graphs shard by rank with no data-path collective; ONE gradient all-reduce per step):
  --scaling strong (default; BASELINE config 4 / SURVEY 8d-e): the global batch stays 4096 and
                   each of the N ranks trains on its contiguous run of 4096/N graphs;
  --scaling weak   every rank trains on its own 4096-graph shard (global batch N*4096).
`python bench.py --gpus N` launches its own N workers (torch.distributed.run as a child process,
before anything touches the GPU); under the driver's torchrun it just joins the group.
--launch auto replays a HIP graph of the step when the per-rank shard is small enough to be
host-bound (< 2048 graphs), else launches eagerly.

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel of the step, timed live
with HIP events on its own stream (`frac` against SURVEY 8d's algorithmic bytes, `real_traffic_frac`
against the HBM bytes the offline PMC passes under profiles/ measured for that kernel);
`cpu_baseline` is the oracle (a port of the reference's pure-PyTorch CPU path) timed on this box's
host cores on a bounded sample.  At N = 1 with the default workload the same process then runs the
other single-GPU BASELINE configs (cfg2 eager + replay, cfg3, cfg5 fp16 / fp32, the 512-graph shard
of the headline) and reports them under `configs` (--no-configs skips them).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)

WORKLOADS = {
    # name: model, n ROI, k, hidden, per-GPU batch, subjects per GPU
    "cfg4-headline-gcn-4096x360-h64": dict(model="gcn", n=360, k=14, hidden=64, batch=4096),
    "cfg2-gcn-512x84-h64": dict(model="gcn", n=84, k=8, hidden=64, batch=512),
    "cfg3-sage-512x360-h128": dict(model="sage", n=360, k=14, hidden=128, batch=512),
    # BASELINE config 5's shape run in fp32 (the reference's arithmetic); dense 1000-ROI graphs,
    # 100k edges each
    "cfg5-gcn-64x1000-h256-fp32": dict(model="gcn", n=1000, k=100, hidden=256, batch=64),
    # BASELINE config 5 as specified: fp16 storage / fp32 accumulate (GCNConnectome(storage="fp16"))
    "cfg5-gcn-64x1000-h256-fp16": dict(model="gcn", n=1000, k=100, hidden=256, batch=64, storage="fp16"),
}


def algorithmic_bytes_per_graph(model: str, n: int, e: int, hidden: int, s: int = 4) -> float:
    """GCN Nn*s*(34H+10) + 48*Ee ; SAGE Nn*s*(37H+10) + 48*Ee (per graph)."""
    c = 34 if model == "gcn" else 37
    return n * s * (c * hidden + 10) + 48.0 * e


def cpu_baseline(model: str, n: int, k: int, hidden: int, budget_s: float = 20.0) -> dict:
    """Time the oracle's full train step (same ATen op sequence as the reference) on the host."""
    from oracle import reference_path as O
    from connectome_gnn_amd.synthetic import generate_packed
    from connectome_gnn_amd.resident import assemble_batch
    sample = 8 if n >= 1000 else (128 if n >= 360 else 512)
    # a 1-GPU box's CPU share is 16 cores; more threads only oversubscribe the ATen loops
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    ds = generate_packed(sample, n, k, seed=42)
    b = assemble_batch(ds, torch.arange(sample))
    ob = O.OBatch(b.node_features, b.edge_index, b.edge_weight, b.batch, b.labels, b.ptr)
    torch.manual_seed(42)
    st = O.require_grad(O.INIT[model](5, hidden))
    opt = torch.optim.Adam([st[kk] for kk in O.param_keys(st)], lr=1e-3, weight_decay=1e-4)
    times = []
    t_end = time.perf_counter() + budget_s
    for i in range(3 + 10):
        t0 = time.perf_counter()
        O.train_step(model, st, ob, opt, 0.3)
        dt = time.perf_counter() - t0
        if i >= 3:
            times.append(dt)
        if time.perf_counter() > t_end and len(times) >= 3:
            break
    times.sort()
    med = times[len(times) // 2]
    cpu_model = "unknown"
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    cpu_model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {"value": sample / med, "unit": "graphs/s", "cores": torch.get_num_threads(),
            "kind": "port", "cpu_model": cpu_model,
            "sample": f"oracle train step (dropout 0.3, Adam), batch {sample}x{n}-ROI, "
                      f"median of {len(times)} steps after 3 warm-up, collate excluded; context only -- "
                      "this figure moves by +-40 % between boxes of the pool (shared host cores)"}


# workload -> (candidate files under profiles/, newest first; batch the passes were taken at;
# kernel-name PREFIXES whose launches make up the dominant kernel).  Prefixes, not full template
# instantiations: a template parameter added to a kernel must not silently null the field.
PMC_FILES = {
    "cfg4-headline-gcn-4096x360-h64": (("r04_headline_pmc_traffic.json", "r03_headline_pmc_traffic.json", "r02_headline_pmc_traffic.json"), 4096,
                                       ("k_gcn_bwd<",)),
    "cfg3-sage-512x360-h128": (("r04_cfg3_pmc_traffic.json", "r03_cfg3_pmc_traffic.json", "r02_cfg3_pmc_traffic.json"), 512, ("k_agg_tiled",)),
    "cfg2-gcn-512x84-h64": (("r04_cfg2_pmc_traffic.json", "r03_cfg2_pmc_traffic.json", "r02_cfg2_pmc_traffic.json"), 512, ("k_gcn_bwd<",)),
    "cfg5-gcn-64x1000-h256-fp16": (("r04_cfg5_fp16_pmc_traffic.json", "r03_cfg5_fp16_pmc_traffic.json", "r02_cfg5_fp16_pmc_traffic.json"), 64,
                                   ("k_dense_agg",)),
    "shard512-gcn-512x360-h64": (("r04_shard512_pmc_traffic.json", "r03_shard512_pmc_traffic.json"), 512, ("k_gcn_bwd<",)),
    "shard512-dp-plumbing-gcn-512x360-h64": (("r04_shard512_pmc_traffic.json", "r03_shard512_pmc_traffic.json"), 512, ("k_gcn_bwd<",)),
    # one aggregation = two launches (dense fragments on the matrix cores, then the remaining edges): SUMMED
    "cfg5-gcn-64x1000-h256-fp32": (("r04_cfg5_fp32_pmc_traffic.json", "r03_cfg5_fp32_pmc_traffic.json"), 64, (("k_band_agg",), ("k_agg_wave_row",))),
}


def pmc_traffic(workload: str, bsz: int):
    """(HBM bytes per launch of the dominant kernel, HBM bytes per step, source) from the committed
    offline PMC passes, or (None, None, None).  A file that exists for this workload but holds
    no kernel matching the prefixes is an ERROR (stale key), not a silent null."""
    if workload not in PMC_FILES or os.environ.get("CGNN_BENCH_COLLECTING_PMC"):
        return None, None, None          # (the counter passes themselves run this script: tools/measure_all.sh)
    files, pmc_bsz, prefixes = PMC_FILES[workload]
    if bsz != pmc_bsz:
        return None, None, None
    for fname in files:
        path = os.path.join(ROOT, "profiles", fname)
        if not os.path.exists(path):
            continue
        doc = json.load(open(path))
        per_launch = 0.0
        for group in (prefixes if isinstance(prefixes[0], tuple) else (prefixes,)):   # groups add up
            hits = [v for k, v in doc["kernels"].items() if any(k.startswith(pf) for pf in group)]
            if not hits:
                raise RuntimeError(f"profiles/{fname}: no kernel starts with any of {group} -- stale PMC_FILES key")
            launches = sum(h["launches"] for h in hits)
            per_launch += sum(h["hbm_bytes_per_launch"] * h["launches"] for h in hits) / launches
        return per_launch, doc.get("hbm_bytes_per_step_library_kernels"), \
            f"profiles/{fname} (offline rocprofv3 --pmc passes, not this run)"
    return None, None, None


def end_to_end(C, ds, model, opt, bsz: int, epochs: int = 6, dataset=None) -> dict:
    """Fresh-batch throughput: Trainer.train_epoch over a ResidentDataLoader
    that re-shuffles every epoch, so on-device assembly, the CSR / blocked-ELL builds of every
    batch AND the step are inside the clock.  `dataset` = the config's own dataset (BASELINE config 4:
    32,768 distinct subjects = 8 steps per epoch), resident in HBM; without it this bench's synthetic
    shard tiled to 4 batches per epoch.  The next batch is built on a side stream (prefetch)."""
    from connectome_gnn_amd.resident import ResidentDataLoader
    from connectome_gnn_amd.synthetic import PackedDataset
    if dataset is not None:
        big = dataset.to(ds.x.device)
        epochs = 3
    else:
        rep = lambda t: t.repeat(4, *([1] * (t.dim() - 1)))
        big = PackedDataset(rep(ds.x), rep(ds.edge_local), rep(ds.edge_weight), rep(ds.labels))
    tr = C.Trainer(model, opt, device=str(ds.x.device), graph=False)

    def timed_epochs(ld, trainer=None, n_epochs=epochs):
        trainer = trainer or tr
        # untimed epochs until the loader's allocation pattern has settled (its side stream grows the
        # caching allocator's pools over the first epochs: single epochs of 3-8 ms per step): stop when an
        # epoch is within 10 % of the fastest so far, after at least 3 and at most 12
        best = float("inf")
        for ep in range(12):
            torch.cuda.synchronize()
            t_ep = time.perf_counter()
            trainer.train_epoch(ld)
            torch.cuda.synchronize()
            t_ep = time.perf_counter() - t_ep
            if os.environ.get("CGNN_BENCH_DEBUG"):
                print(f"[bench] warm-up epoch {ep}: {t_ep * 1e3 / max(len(ld), 1):.3f} ms/step", file=sys.stderr, flush=True)
            if ep >= 2 and t_ep <= 1.1 * best:
                break
            best = min(best, t_ep)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n_epochs):
            t_ep = time.perf_counter()
            trainer.train_epoch(ld)
            if os.environ.get("CGNN_BENCH_DEBUG"):
                torch.cuda.synchronize()
                print(f"[bench] timed epoch: {(time.perf_counter() - t_ep) * 1e3 / max(len(ld), 1):.3f} ms/step", file=sys.stderr, flush=True)
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    ld = ResidentDataLoader(big, batch_size=bsz, shuffle=True, prefetch=True, prepare=model.prepare_batch)
    dt = timed_epochs(ld)
    steps = epochs * len(ld)
    nbytes = lambda *ts: int(sum(t.numel() * t.element_size() for t in ts))
    out = {"graphs_per_s": steps * bsz / dt, "ms_per_step": dt / steps * 1e3, "steps": steps,
           "subjects": big.num_subjects, "steps_per_epoch": len(ld),
           "dataset_hbm_bytes": nbytes(big.x, big.edge_local, big.edge_weight, big.labels),
           "what": "assemble + CSR/blocked-ELL build + step per fresh shuffled batch, prefetch on a "
                   "side stream, Trainer.train_epoch API, eager launches"}
    # the same loop with the batches composed once and only their ORDER re-drawn every epoch
    # (ResidentDataLoader(shuffle="batches", cache_batches=True)): per-batch work is paid once
    ld2 = ResidentDataLoader(big, batch_size=bsz, shuffle="batches", cache_batches=True,
                             prepare=model.prepare_batch)
    dt2 = timed_epochs(ld2)
    out["cached_batches"] = {"graphs_per_s": steps * bsz / dt2, "ms_per_step": dt2 / steps * 1e3,
                             "what": "same Trainer loop, fixed batch composition (order shuffled), "
                                     "structure cached per batch"}
    # fresh composition every epoch again, but the structure of a batch gathered from the
    # per-SUBJECT cache (structure_cache.py; one graph per tile, per-tile GCN path)
    n = int(big.x.shape[1])
    if n <= 384 and getattr(model, "_fused_kind", None) == "tile":
        from connectome_gnn_amd.structure_cache import SubjectStructureCache
        torch.cuda.synchronize()
        t_c = time.perf_counter()
        cache = SubjectStructureCache(big)                 # one per dataset, shared by the loaders below
        torch.cuda.synchronize()
        fam = cache.family("gcn")
        out["subject_cache_build"] = {
            "seconds": time.perf_counter() - t_c,
            "hbm_bytes": nbytes(fam.ent_dst, fam.ent_src, fam.blk_off_dst, fam.blk_off_src, fam.norm),
            "what": "blocked-ELL entries (both orderings), block offsets and dis of every subject, built once"}
        ld3 = ResidentDataLoader(big, batch_size=bsz, shuffle=True, prefetch=True, prepare=model.prepare_batch)
        ld3.structure_cache = cache
        dt3 = timed_epochs(ld3)
        out["subject_cache"] = {"graphs_per_s": steps * bsz / dt3, "ms_per_step": dt3 / steps * 1e3,
                                "what": "fresh shuffled batch every step; blocked-ELL / dis of every "
                                        "subject built once, a batch's structure = three gathers on the side stream"}
        # ... and with Trainer(graph=True): one captured step per batch size, the batch assembled INSIDE
        # the graph from its subject ids (graphed.GraphedResidentStep) -- per-epoch reshuffling at replay
        # speed; also at one rank's 512-graph share of the batch (8-GPU strong scaling with a real loader)
        if getattr(opt, "defaults", {}).get("capturable", False):
            trg = C.Trainer(model, opt, device=str(ds.x.device), graph=True)
            for tag, b2 in (("subject_cache_graph", bsz), ("subject_cache_graph_512", 512)):
                if b2 > bsz:
                    continue
                ld4 = ResidentDataLoader(big, batch_size=b2, shuffle=True, prefetch=True, prepare=model.prepare_batch)
                ld4.structure_cache = cache
                ne = epochs if b2 == bsz else (1 if dataset is not None else 2)
                dt4 = timed_epochs(ld4, trg, ne)
                st4 = ne * len(ld4)
                out[tag] = {"graphs_per_s": st4 * b2 / dt4, "ms_per_step": dt4 / st4 * 1e3, "batch": b2,
                            "steps_per_epoch": len(ld4),
                            "what": "fresh shuffled batch every step through Trainer(graph=True): HIP-graph "
                                    "replay with the batch assembled inside the graph from its subject ids"}
            trg.clear_graphs()
    return out


def trainer_replay_record(args, name: str, dev, label: str) -> dict:
    """BASELINE config 2 the way a user of the drop-in API runs it: `Trainer(graph=True).train_epoch` over
    a `ResidentDataLoader(shuffle=True, structure_cache=True)` -- every epoch re-draws every batch
    (reference graph.py:190-197), the batch is assembled inside ONE captured step per batch size
    (graphed.GraphedResidentStep), nothing but the subject ids goes to the device per step.  Timed
    end to end: loader, id copies, replays, the per-epoch loss read-back."""
    import connectome_gnn_amd as C
    from connectome_gnn_amd.optim import Adam
    from connectome_gnn_amd.resident import ResidentDataLoader
    from connectome_gnn_amd.synthetic import generate_packed
    wl = WORKLOADS[name]
    n, k, hidden, bsz = wl["n"], wl["k"], wl["hidden"], wl["batch"]
    ds = generate_packed(8 * bsz, n, k, seed=42).to(dev)       # cfg2 / cfg3: 4096 subjects, batches of 512
    torch.manual_seed(42)
    cls = C.GCNConnectome if wl["model"] == "gcn" else C.GraphSAGEConnectome
    model = cls(5, hidden, 2, 3, 0.3).to(dev).train()
    opt = Adam(model.parameters(), lr=1e-3, weight_decay=1e-4)
    tr = C.Trainer(model, opt, device=str(dev), graph=True)
    ld = ResidentDataLoader(ds, batch_size=bsz, shuffle=True, structure_cache=True, prepare=model.prepare_batch)
    for _ in range(2):
        tr.train_epoch(ld)
    epochs = max(1, -(-args.steps // len(ld)))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(epochs):
        last = tr.train_epoch(ld)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    steps = epochs * len(ld)
    bpg = algorithmic_bytes_per_graph(wl["model"], n, n * k, hidden)
    gps = steps * bsz / dt
    tr.clear_graphs()
    return {"workload": label, "launch": "Trainer(graph=True).train_epoch, fresh shuffled batches (hip-graph replay, "
                                         "batch assembled inside the graph from a per-subject structure cache)",
            "dtype": "f32", "graphs_per_gpu": bsz, "impl": getattr(model, "impl_used", None),
            "ms_per_step": dt / steps * 1e3, "value": gps, "unit": "graphs/s", "steps": steps,
            "step_algorithmic": {"frac": bpg * gps / (HBM_PEAK_GBS * 1e9), "bytes_per_graph": bpg,
                                 "real_hbm_bytes_per_step": None, "real_traffic_frac": None},
            "roofline": None, "final_loss": last}


def demo_record(dev) -> dict:
    """BASELINE config 1: the reference's demo experiment (examples/demo.py = /root/reference/examples/demo.py:35-134
    on this package: 300 x 84-ROI subjects, GCN + GraphSAGE hidden 64, batch 16, Adam, 30 epochs, patience 8)
    through the drop-in API with its defaults -- list-backed ConnectomeDataLoader, Trainer(model, torch.optim.Adam).
    Wall seconds of the whole script incl. data generation (reference on 8 CPU cores: 11.2 s)."""
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    import demo
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = demo.run(device=str(dev), epochs=30, verbose=False)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"workload": "cfg1-demo-300x84-gcn+sage-h64-b16", "launch": "examples/demo.py, Trainer defaults",
            "wall_seconds": dt, "unit": "s", "higher_is_better": False,
            "test_accuracy": {k: v["test"]["accuracy"] for k, v in res.items()},
            "epochs_run": {k: len(v["history"]["train_loss"]) for k, v in res.items()},
            "impl": {k: v["impl"] for k, v in res.items()}}


def plain_loader_record(args, name: str, dev, label: str) -> dict:
    """A config the way the UNCHANGED reference script runs it: a Python list of ConnectomeGraph on the host,
    `ConnectomeDataLoader(graphs, batch_size, shuffle=True)` (reference graph.py:174-197) and
    `Trainer(model, torch.optim.Adam(...), device)` with every default (train.py:19-54).  The Trainer packs
    the list into HBM on first sight and replays a captured step per batch size (train.py module docstring);
    the one-off packing is reported beside the steady-state step."""
    import connectome_gnn_amd as C
    from connectome_gnn_amd.synthetic import generate_packed
    wl = WORKLOADS[name]
    n, k, hidden, bsz = wl["n"], wl["k"], wl["hidden"], wl["batch"]
    host = generate_packed(8 * bsz, n, k, seed=42)
    graphs = [host.graph(i) for i in range(host.num_subjects)]
    torch.manual_seed(42)
    cls = C.GCNConnectome if wl["model"] == "gcn" else C.GraphSAGEConnectome
    model = cls(5, hidden, 2, 3, 0.3)
    tr = C.Trainer(model, torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-4), device=str(dev))
    ld = C.ConnectomeDataLoader(graphs, batch_size=bsz, shuffle=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr.train_epoch(ld)                       # packs, builds the subject cache, captures
    torch.cuda.synchronize()
    first = time.perf_counter() - t0
    tr.train_epoch(ld)
    epochs = max(1, -(-args.steps // len(ld)))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(epochs):
        last = tr.train_epoch(ld)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    steps = epochs * len(ld)
    gps = steps * bsz / dt
    bpg = algorithmic_bytes_per_graph(wl["model"], n, n * k, hidden)
    rec = {"workload": label, "launch": "ConnectomeDataLoader(list) + Trainer(model, torch.optim.Adam) defaults "
                                        f"({'hip-graph replay' if tr.graph else 'eager'})",
           "dtype": "f32", "graphs_per_gpu": bsz, "impl": getattr(model, "impl_used", None),
           "ms_per_step": dt / steps * 1e3, "value": gps, "unit": "graphs/s", "steps": steps,
           "first_epoch_seconds": first,
           "step_algorithmic": {"frac": bpg * gps / (HBM_PEAK_GBS * 1e9), "bytes_per_graph": bpg,
                                "real_hbm_bytes_per_step": None, "real_traffic_frac": None},
           "roofline": None, "final_loss": last}
    tr.clear_graphs()
    return rec


def inference_record(name: str, dev, label: str, dataset=None) -> dict:
    """N4: `Trainer.evaluate` (reference train.py:56-74) over a device-resident dataset, eval-mode
    BatchNorm, no dropout; graphs/s incl. the loader, the loss / hit tallies and the one read-back."""
    import connectome_gnn_amd as C
    from connectome_gnn_amd.optim import Adam
    from connectome_gnn_amd.resident import ResidentDataLoader
    from connectome_gnn_amd.synthetic import generate_packed
    wl = WORKLOADS[name]
    n, k, hidden, bsz = wl["n"], wl["k"], wl["hidden"], wl["batch"]
    ds = (dataset if dataset is not None else generate_packed(8 * bsz, n, k, seed=42)).to(dev)
    torch.manual_seed(42)
    cls = C.GCNConnectome if wl["model"] == "gcn" else C.GraphSAGEConnectome
    model = cls(5, hidden, 2, 3, 0.3).to(dev)
    use_graph = bsz < 2048          # (as --launch auto: small batches are host-bound when launched eagerly)
    tr = C.Trainer(model, Adam(model.parameters(), lr=1e-3), device=str(dev), graph=use_graph)
    ld = ResidentDataLoader(ds, batch_size=bsz, shuffle=False, structure_cache=True)
    for _ in range(2):
        tr.evaluate(ld)
    reps = max(1, 24 // len(ld))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        ev = tr.evaluate(ld)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"workload": label, "launch": "Trainer.evaluate over ResidentDataLoader(structure_cache=True), "
                                         + ("hip-graph replay (graphed.GraphedEvalStep)" if use_graph else "eager"),
            "dtype": "f32", "graphs_per_gpu": bsz, "impl": getattr(model, "impl_used", None),
            "ms_per_batch": dt / (reps * len(ld)) * 1e3, "value": reps * ds.num_subjects / dt, "unit": "graphs/s",
            "subjects": ds.num_subjects, "accuracy": ev["accuracy"]}


def spawn_workers(n: int) -> int:
    """`python bench.py --gpus N` outside torchrun: start N ranks as a CHILD torch.distributed.run
    (never an exec: this process may not be replaced once anything has initialised the GPU, and
    here nothing has) and hand its exit code back."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def note(rank: int, msg: str) -> None:
    if rank == 0:
        print(f"[bench] {msg}", file=sys.stderr, flush=True)


def plan_run(scaling: str, launch: str, gbatch: int, rank: int, world: int) -> dict:
    """The rank bookkeeping of one run, free of any GPU work (tests/test_host_logic.py drives it for --gpus 8):
    which graphs this rank trains on, what the global batch is, and how the step is launched.
    strong: the global batch `gbatch` is cut into contiguous runs, sizes differing by at most one
    graph; weak: every rank trains on its own `gbatch` graphs.  launch "auto" replays a HIP graph when the
    per-rank shard is host-bound (< 2048 graphs), else launches eagerly."""
    if scaling == "strong":
        from connectome_gnn_amd.graph import shard_slice
        shard_sizes = [len(shard_slice(list(range(gbatch)), r, world)) for r in range(world)]
        bsz, global_batch = shard_sizes[rank], gbatch
        if min(shard_sizes) == 0:
            raise SystemExit(f"global batch {gbatch} < world size {world}")
    else:
        shard_sizes = [gbatch] * world
        bsz, global_batch = gbatch, gbatch * world
    if launch == "auto":
        launch = "graph" if bsz < 2048 else "eager"
    return {"shard_sizes": shard_sizes, "graphs_this_rank": bsz, "global_batch": global_batch,
            "equal_shards": len(set(shard_sizes)) == 1, "launch": launch,
            "local_graphs": None if len(set(shard_sizes)) == 1 else bsz}


def parallel_config(world: int, backend, scaling: str, sync_bn: bool, graphed: bool, collectives: str,
                    launch_note) -> dict:
    """The `config` keys that say what ran across ranks (backend "nccl" IS RCCL on ROCm)."""
    return {"launch": ("hip-graph replay" + (f" ({collectives} all-reduce)" if world > 1 else "")) if graphed else "eager",
            "launch_note": launch_note,
            "bn": "sync" if (world > 1 and sync_bn) else "per-rank",
            "backend": ("rccl" if backend == "nccl" else backend), "rccl_ranks": world if backend == "nccl" else 0,
            "parallelism": f"graph-sharded dp{world}"}


def projection_8gpu(headline_ms: float, shard_ms: float, gbatch: int = 4096) -> dict:
    """NOT a measurement: what the 1-GPU records imply for 8-GPU strong scaling of the headline batch -- each
    rank's 512-graph step as timed here on one GPU plus ONE gradient all-reduce of 45 KB (latency-bound over
    xGMI; 30-50 us assumed, RCCL has not run on this build's hardware).  Reported under `projection`, never as
    `value`; the driver's SCALE_rNN.json is the measurement when a node exists."""
    out = {"what": "projection from 1-GPU timings, not measured", "n_gpus": 8, "global_batch": gbatch,
           "shard_ms_per_step_measured_1gpu": shard_ms, "allreduce_ms_assumed": [0.03, 0.05],
           "headline_ms_per_step_measured_1gpu": headline_ms}
    gps = [gbatch / ((shard_ms + a) * 1e-3) for a in out["allreduce_ms_assumed"]]
    out["graphs_per_s"] = gps
    out["speedup_vs_1gpu"] = [g / (gbatch / (headline_ms * 1e-3)) for g in gps]
    return out


def run_workload(args, name: str, rank: int, world: int, dev, *, launch: str, batch: int = 0,
                 label: str = None, extras: bool = True, e2e_dataset=None, dp_plumbing: bool = False) -> dict:
    """Warm up, time `args.steps` steps of workload `name` and return the JSON record (on every
    rank; only rank 0's is printed).  `label` names the record when it differs from the workload
    key (the 512-graph shard of the headline)."""
    import connectome_gnn_amd as C
    from connectome_gnn_amd import _lib, dist as cdist, ops as cops
    from connectome_gnn_amd.resident import assemble_batch
    from connectome_gnn_amd.synthetic import generate_packed

    label = label or name
    wl = dict(WORKLOADS[name])
    if batch:
        wl["batch"] = batch
    model_kind, n, k, hidden, gbatch = wl["model"], wl["n"], wl["k"], wl["hidden"], wl["batch"]
    e = n * k
    plan = plan_run(args.scaling, launch, gbatch, rank, world)
    shard_sizes, bsz, global_batch = plan["shard_sizes"], plan["graphs_this_rank"], plan["global_batch"]
    equal_shards, launch, use_graph = plan["equal_shards"], plan["launch"], plan["launch"] == "graph"

    # ---- data: this rank's shard of the synthetic dataset, resident in HBM ------------------
    ds = generate_packed(bsz, n, k, seed=42 + rank).to(dev)
    if args.node_order == "degree":
        ds = ds.relabel_by_degree()
    g = torch.Generator().manual_seed(1234 + rank)
    batches = []
    for _ in range(max(1, args.nbuf)):
        b = assemble_batch(ds, torch.randperm(bsz, generator=g))
        b.structure()                       # CSR build = collate-time work, outside the step
        if os.environ.get("CGNN_DIAG_ZERO_FEATURES"):   # diagnostic only (clock held without data
            b.node_features.zero_()                     # toggling, MI355X_MICROARCH 'DVFS give-back')
        batches.append(b)

    torch.manual_seed(42)
    cls = C.GCNConnectome if model_kind == "gcn" else C.GraphSAGEConnectome
    kw = {} if args.impl == "auto" else {"impl": args.impl}
    storage = wl.get("storage", "fp32")
    if storage != "fp32":
        kw["storage"] = storage
    elem = 2 if storage == "fp16" else 4
    model = cls(5, hidden, 2, 3, 0.3, **kw).to(dev).train()
    if world > 1:
        cdist.broadcast_parameters(model)
        if args.sync_bn:
            model = cdist.convert_sync_batchnorm(model)
    # (--dp-plumbing: one rank with the data-parallel gradient path in place -- gradients land in the flat
    # all-reduce buffer, the exchange itself is a no-op at world 1: what a rank's step costs besides the wire)
    dp_plumbing = dp_plumbing or getattr(args, "dp_plumbing", False)
    sync = cdist.GradSync(model.parameters()) if (world > 1 or dp_plumbing) else None
    collectives = args.collectives
    if use_graph and world > 1 and args.sync_bn:
        collectives = "captured"          # sync-BN exchanges sit inside forward/backward
    if args.optimizer == "torch":
        opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-4, fused=True,
                               capturable=use_graph)
    else:
        from connectome_gnn_amd.optim import Adam      # torch.optim.Adam's update as one launch
        opt = Adam(model.parameters(), lr=1e-3, weight_decay=1e-4)
    loss_fn = cops.CrossEntropyLoss()        # the Trainer's criterion: CrossEntropyLoss defaults
    local_graphs = plan["local_graphs"]

    def eager_step(i: int):
        b = batches[i % len(batches)]
        if sync is not None:
            sync.zero_grad()                 # grads are views of the flat all-reduce buffer
        else:
            opt.zero_grad(set_to_none=True)
        loss = cops.model_loss(model, loss_fn, b)       # (= loss_fn(model(b), b.labels), as Trainer.train_step)
        cops.backward_unit(loss)
        if sync is not None:
            sync(local_graphs=local_graphs)
        opt.step()
        return loss

    graphed, launch_note = None, None
    if use_graph:
        # Capture is attempted in THIS process and abandoned in this process: if it fails on any
        # rank (nobody could rehearse capture under RCCL on an 8-GPU node), every rank drops to
        # eager launches together -- the run still produces its number.
        from connectome_gnn_amd.graphed import GraphedTrainStep
        ok = True
        try:
            graphed = [GraphedTrainStep(model, opt, b, loss_fn, grad_sync=sync, collectives=collectives,
                                        local_graphs=local_graphs) for b in batches]
        except Exception as exc:             # noqa: BLE001 -- any capture failure means "go eager"
            ok = False
            launch_note = f"graph capture failed on rank {rank} ({type(exc).__name__}: {exc}); eager launches"
            print(f"[bench rank {rank}] {launch_note}", file=sys.stderr, flush=True)
            torch.cuda.synchronize()
        if world > 1 and not cdist.agree(ok, dev):
            ok = False
            launch_note = launch_note or "graph capture failed on another rank; eager launches"
        if not ok:
            graphed = None
            for prm in model.parameters():   # a half-captured step may have left pool-owned grads
                prm.grad = None
            if sync is not None:
                sync.zero_grad()

    def step(i: int):
        if graphed is not None:
            return graphed[i % len(graphed)]()
        return eager_step(i)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for bt in batches:
        model.prepare_batch(bt, reuse=True)   # resident batches every step trains on: amortised structure work
    for i in range(args.warmup):
        step(i)
    impl_used = getattr(model, "impl_used", "layered")
    # dominant kernel of the step and its ALGORITHMIC bytes per launch (SURVEY 8d per-unit
    # figures): fused path -> the per-layer backward kernel: BN-backward 5*Nn*H*s +
    # Nn*s*(H + 2*F_l) + 8*Ee; layered path -> the aggregate: 2*Nn*F*s + 8*Ee + 4*(Nn+1).
    agg_bytes = lambda nn_, ee: 2.0 * nn_ * hidden * 4 + 8.0 * ee + 4.0 * (nn_ + 1)
    candidates = {
        "cgnn_gcn_fused_bwd": lambda nn_, ee: nn_ * 4.0 * (5 * hidden + hidden + 2 * hidden) + 8.0 * ee,
        f"cgnn_aggregate_tiled_f32[F={hidden}]": agg_bytes,     # LDS-tiled aggregate (wide layers)
        f"cgnn_aggregate_f32[F={hidden}]": agg_bytes,           # gather form (graphs > 384 nodes)
        # ... with the operator's dense fragments on the matrix cores: two launches per aggregation, timed
        # and priced as ONE (the sparse operator's bytes; the fragments' stored operands are not "algorithmic")
        f"cgnn_band_aggregate_f32+cgnn_aggregate_acc_f32[F={hidden}]": agg_bytes,
        # fp16 storage: the dense per-graph aggregate, priced at the SPARSE operator's bytes (s = 2)
        f"cgnn_dense_aggregate_f16[F={hidden}]": lambda nn_, ee: 2.0 * nn_ * hidden * 2 + 8.0 * ee + 4.0 * (nn_ + 1),
        f"cgnn_dense_aggregate_c16[F={hidden}]": lambda nn_, ee: 2.0 * nn_ * hidden * 2 + 8.0 * ee + 4.0 * (nn_ + 1),
    }
    fused_kind = getattr(model, "_fused_kind", None) if impl_used == "fused" else None
    if fused_kind == "half":
        from connectome_gnn_amd import gcn_half_path, ops as _ops
        packed = isinstance(gcn_half_path.dense_operators(batches[0].structure())[0], _ops.DensePack)
        dom = f"cgnn_dense_aggregate_{'c16' if packed else 'f16'}[F={hidden}]"
    elif fused_kind == "tile":
        dom = "cgnn_gcn_fused_bwd"
    elif batches[0].structure().tiled_ok(hidden):
        dom = f"cgnn_aggregate_tiled_f32[F={hidden}]"
    elif any(v is not None for v in batches[0].structure().__dict__.get("_band_ops", {}).values()):
        dom = f"cgnn_band_aggregate_f32+cgnn_aggregate_acc_f32[F={hidden}]"
    else:
        dom = f"cgnn_aggregate_f32[F={hidden}]"
    dom_bytes_fn = candidates[dom]
    if graphed is None:
        _lib.TIMER = _lib.KernelTimer([dom])
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = step(args.warmup + i)
    fence()
    dt = time.perf_counter() - t0
    if graphed is not None:          # kernels inside a graph cannot be bracketed: time them eagerly
        _lib.TIMER = _lib.KernelTimer([dom])
        for i in range(4):
            eager_step(i)
        torch.cuda.synchronize()
    timer, _lib.TIMER = _lib.TIMER, None
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t)
    final_loss = float(loss.detach())
    loss = None     # drop the last eager autograd graph: its AccumulateGrad nodes are tied to this stream,
                    # and a later capture of the same parameters on another stream must not meet them

    graphs_per_s = global_batch * args.steps / dt
    bpg = algorithmic_bytes_per_graph(model_kind, n, e, hidden, elem)
    kms = timer.ms(dom)
    nn_, ee = bsz * n, bsz * e
    dom_bytes = dom_bytes_fn(nn_, ee)
    avg_ms = sum(kms) / max(len(kms), 1)
    achieved = dom_bytes / (avg_ms * 1e-3) / 1e9 if kms else 0.0
    # HBM bytes per launch of that kernel: NOT measured in this run -- read from the PMC passes
    # kept under profiles/ (collected offline with rocprofv3 --pmc on this same command, FETCH
    # and WRITE in separate passes, corrected as MI355X_MICROARCH.md prescribes); null when no
    # pass exists for this workload / batch.  `traffic_source` names the file.
    traffic, step_traffic, traffic_source = pmc_traffic(label, bsz)
    real_frac = (traffic / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if (traffic and kms) else None
    backend = torch.distributed.get_backend() if world > 1 else None
    out = {
        "metric": "training graphs/sec, 3-layer GCN, batch=4096x360-ROI connectomes"
        if label.startswith("cfg4") else f"training graphs/sec, {label}",
        "value": graphs_per_s, "unit": "graphs/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
        "dtype": "f16" if storage == "fp16" else "f32",
        "data": "synthetic",
        "config": {"workload": label, "model": model_kind, "rois": n, "ws_k": k,
                   "edges_per_graph": e, "hidden": hidden, "layers": 3, "node_order": args.node_order,
                   "graphs_per_gpu": bsz, "global_batch": global_batch, "shard_sizes": shard_sizes,
                   "dropout": 0.3,
                   "optimizer": "Adam lr1e-3 wd1e-4" + (" (torch fused)" if args.optimizer == "torch" else " (optim.Adam)"), "impl": impl_used,
                   **parallel_config(world, backend, args.scaling, args.sync_bn, graphed is not None, collectives,
                                     launch_note)},
        "step_algorithmic": {"bytes_per_graph": bpg,
                             "GBps": bpg * graphs_per_s / world / 1e9,
                             "frac_of_hbm_peak": bpg * graphs_per_s / world / (HBM_PEAK_GBS * 1e9),
                             "real_hbm_bytes_per_step": step_traffic,
                             "real_traffic_frac": (step_traffic / (dt / args.steps) / (HBM_PEAK_GBS * 1e9))
                             if step_traffic else None},
        "roofline": {"bound": "hbm", "kernel": dom, "launches_timed": len(kms),
                     "avg_ms": avg_ms, "achieved": achieved, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "real_traffic_frac": real_frac,
                     "traffic_source": traffic_source,
                     "algorithmic_bytes_per_launch": dom_bytes},
        "final_loss": final_loss,
    }
    if extras and rank == 0 and world == 1 and not args.no_end_to_end and graphed is None \
            and fused_kind == "tile" and args.node_order == "dataset":
        # the same step on the same graphs with every subject's nodes renumbered by degree
        # (PackedDataset.relabel_by_degree: a one-off preprocessing of the dataset; the models
        # are invariant under it): less blocked-ELL padding for the tile kernels to walk
        ds2 = ds.relabel_by_degree()
        g2 = torch.Generator().manual_seed(1234 + rank)
        batches[:] = [assemble_batch(ds2, torch.randperm(bsz, generator=g2)) for _ in batches]
        for b2 in batches:
            b2.structure()
            model.prepare_batch(b2)
        for i in range(args.warmup):
            eager_step(i)
        fence()
        t1 = time.perf_counter()
        for i in range(args.steps):
            eager_step(i)
        fence()
        dt2 = time.perf_counter() - t1
        out["relabelled_by_degree"] = {"graphs_per_s": global_batch * args.steps / dt2,
                                       "ms_per_step": dt2 / args.steps * 1e3,
                                       "what": "same step, dataset preprocessed with "
                                               "PackedDataset.relabel_by_degree()"}
    if extras and rank == 0 and world == 1 and not args.no_end_to_end and n <= 384:
        out["end_to_end"] = end_to_end(C, ds, model, opt, bsz, dataset=e2e_dataset)
    return out


# The other single-GPU BASELINE configs, measured in the same process after the headline and
# reported under "configs" (VERDICT r2 #1): (record label, workload key, launch mode, batch override)
EXTRA_CONFIGS = (
    ("cfg2-gcn-512x84-h64", "cfg2-gcn-512x84-h64", "eager", 0),
    ("cfg2-gcn-512x84-h64", "cfg2-gcn-512x84-h64", "graph", 0),
    # the same config through the drop-in Trainer with per-epoch reshuffling (VERDICT r2 missing #4)
    ("cfg2-gcn-512x84-h64", "cfg2-gcn-512x84-h64", "trainer", 0),
    # ... and through the reference script's own plumbing: list of graphs, ConnectomeDataLoader, Trainer defaults
    ("cfg2-gcn-512x84-h64", "cfg2-gcn-512x84-h64", "plain", 0),
    ("cfg1-demo", "cfg2-gcn-512x84-h64", "demo", 0),
    ("cfg3-sage-512x360-h128", "cfg3-sage-512x360-h128", "graph", 0),
    ("cfg3-sage-512x360-h128", "cfg3-sage-512x360-h128", "trainer", 0),
    ("cfg5-gcn-64x1000-h256-fp16", "cfg5-gcn-64x1000-h256-fp16", "graph", 0),
    ("cfg5-gcn-64x1000-h256-fp32", "cfg5-gcn-64x1000-h256-fp32", "graph", 0),
    # one rank's share of the headline batch at 8 GPUs (strong scaling), for the >= 6x projection
    ("shard512-gcn-512x360-h64", "cfg4-headline-gcn-4096x360-h64", "graph", 512),
    # ... and the same shard with the data-parallel gradient path in place (dist.GradSync: gradients written into
    # the flat all-reduce buffer; at world 1 the exchange itself is a no-op): a rank's step minus the wire
    ("shard512-dp-plumbing-gcn-512x360-h64", "cfg4-headline-gcn-4096x360-h64", "graph-dp", 512),
)


def summarise(rec: dict) -> dict:
    """The per-config entry of `configs`: what the judge asked for, nothing model-sized."""
    r = rec["roofline"]
    return {"workload": rec["config"]["workload"], "launch": rec["config"]["launch"], "dtype": rec["dtype"],
            "graphs_per_gpu": rec["config"]["graphs_per_gpu"], "impl": rec["config"]["impl"],
            "ms_per_step": rec["ms_per_step"], "value": rec["value"], "unit": rec["unit"],
            "step_algorithmic": {"frac": rec["step_algorithmic"]["frac_of_hbm_peak"],
                                 "bytes_per_graph": rec["step_algorithmic"]["bytes_per_graph"],
                                 "real_hbm_bytes_per_step": rec["step_algorithmic"]["real_hbm_bytes_per_step"],
                                 "real_traffic_frac": rec["step_algorithmic"]["real_traffic_frac"]},
            "roofline": {"kernel": r["kernel"], "avg_ms": r["avg_ms"], "frac": r["frac"],
                         "traffic": r["traffic"], "real_traffic_frac": r["real_traffic_frac"],
                         "traffic_source": r["traffic_source"]},
            "final_loss": rec["final_loss"]}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="cfg4-headline-gcn-4096x360-h64", choices=list(WORKLOADS))
    ap.add_argument("--impl", default=os.environ.get("CGNN_IMPL", "auto"))
    ap.add_argument("--batch", type=int, default=0, help="override per-GPU batch")
    ap.add_argument("--nbuf", type=int, default=2, help="distinct resident batches cycled through")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sync-bn", action="store_true",
                    help="full-batch BN statistics across ranks (default OFF: per-rank BN, what stock DDP does)")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="strong: global batch fixed, split over ranks; weak: fixed batch per rank")
    ap.add_argument("--launch", default="auto", choices=["auto", "eager", "graph"],
                    help="graph: replay a HIP graph of the whole step (kernel timing then comes from a "
                         "short eager pass after the timed region); auto: graph below 2048 graphs/rank")
    ap.add_argument("--graph", action="store_true", help="same as --launch graph")
    ap.add_argument("--collectives", default="split", choices=["split", "captured"],
                    help="graph launch at N > 1: all-reduce between two graphs, or captured inside one")
    ap.add_argument("--node-order", default="dataset", choices=["dataset", "degree"],
                    help="degree: PackedDataset.relabel_by_degree() before batching (an invariance of the models)")
    ap.add_argument("--optimizer", default="cgnn", choices=["cgnn", "torch"],
                    help="cgnn: connectome_gnn_amd.optim.Adam (one launch); torch: torch.optim.Adam(fused=True)")
    ap.add_argument("--no-end-to-end", action="store_true",
                    help="skip the fresh-batch (assemble + structure build + step) measurement")
    ap.add_argument("--no-configs", action="store_true",
                    help="headline only: skip the other single-GPU BASELINE configs (cfg2/3/5, 512-graph shard)")
    ap.add_argument("--backend", default=None, help="torch.distributed backend (default nccl = RCCL)")
    ap.add_argument("--subjects", type=int, default=32768,
                    help="subjects of the headline config's dataset for the end-to-end (loader-inclusive) "
                         "measurement: BASELINE config 4 says 32768 (8 steps of 4096 per epoch on one GPU)")
    ap.add_argument("--dp-plumbing", action="store_true",
                    help="single rank with dist.GradSync in place (flat gradient buffer, exchange a no-op): the "
                         "per-rank step of a data-parallel run without the wire")
    ap.add_argument("--one-device", action="store_true",
                    help="rehearsal only: every rank on cuda:0 (use with --backend gloo)")
    args = ap.parse_args()
    if args.graph:
        args.launch = "graph"
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_workers(args.gpus))      # nothing has touched the GPU yet

    from connectome_gnn_amd import dist as cdist

    # BASELINE config 4's dataset (32,768 x 360-ROI subjects, 3.5 GB): generated on the host cores by forked
    # workers NOW, before this process touches the GPU (a fork afterwards would share the device handles)
    e2e_dataset = None
    headline_default = args.workload.startswith("cfg4") and not args.batch and args.impl == "auto"
    if args.gpus == 1 and headline_default and not args.no_end_to_end and args.subjects > 4096:
        from connectome_gnn_amd.synthetic import generate_packed
        wl0 = WORKLOADS[args.workload]
        t_gen = time.perf_counter()
        e2e_dataset = generate_packed(args.subjects, wl0["n"], wl0["k"], seed=4242,
                                      workers=min(16, os.cpu_count() or 1))
        note(0, f"generated {args.subjects} x {wl0['n']}-ROI subjects in {time.perf_counter() - t_gen:.1f} s")

    if args.one_device:
        os.environ["LOCAL_RANK"] = "0"
    rank, world, local = cdist.init_from_env(args.backend)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if world > 1:
        assert torch.distributed.get_world_size() == args.gpus, "process group size != --gpus"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    out = run_workload(args, args.workload, rank, world, dev, launch=args.launch, batch=args.batch,
                       e2e_dataset=e2e_dataset)
    if world == 1 and headline_default and not args.no_configs:
        out["inference"] = []
        for label, key, data in (("cfg4-headline-gcn-4096x360-h64", args.workload, e2e_dataset),
                                 ("cfg2-gcn-512x84-h64", "cfg2-gcn-512x84-h64", None)):
            try:
                out["inference"].append(inference_record(key, dev, label, data))
            except Exception as exc:         # noqa: BLE001
                out["inference"].append({"workload": label, "error": f"{type(exc).__name__}: {exc}"})
                torch.cuda.synchronize()
        e2e_dataset = None
        import gc
        out["configs"] = []
        for label, key, launch, batch in EXTRA_CONFIGS:
            gc.collect()
            torch.cuda.empty_cache()
            note(rank, f"config {label} ({launch})")
            try:
                if launch == "trainer":
                    out["configs"].append(trainer_replay_record(args, key, dev, label))
                    continue
                if launch == "plain":
                    out["configs"].append(plain_loader_record(args, key, dev, label))
                    continue
                if launch == "demo":
                    out["configs"].append(demo_record(dev))
                    continue
                rec = run_workload(args, key, rank, world, dev, launch=launch.replace("-dp", ""), batch=batch,
                                   label=label, extras=False, dp_plumbing=launch.endswith("-dp"))
                out["configs"].append(summarise(rec))
            except Exception as exc:         # noqa: BLE001 -- one config must not cost the headline line
                out["configs"].append({"workload": label, "launch": launch, "error": f"{type(exc).__name__}: {exc}"})
                torch.cuda.synchronize()
    if rank == 0 and world == 1 and "configs" in out:
        shard = [c for c in out["configs"] if c.get("workload", "").startswith("shard512-dp") and "ms_per_step" in c] \
            or [c for c in out["configs"] if c.get("workload", "").startswith("shard512") and "ms_per_step" in c]
        if shard:
            out["projection"] = projection_8gpu(out["ms_per_step"], shard[0]["ms_per_step"])
            out["projection"]["shard_record"] = shard[0]["workload"]
    if rank == 0:
        if not args.no_cpu_baseline and world == 1:
            wl = WORKLOADS[args.workload]
            out["cpu_baseline"] = cpu_baseline(wl["model"], wl["n"], wl["k"], wl["hidden"])
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
