"""N > 1 parity on the GPU box.  On a one-GPU box two processes share the GPU and talk over gloo
(RCCL refuses two ranks on one device; the collectives' semantics are the same); where two
devices are visible the same test also runs over backend "nccl" (= RCCL, ReduceOp.AVG) with one
rank per device.  With SyncBatchNorm the graph-sharded run must reproduce the single-process
full-batch run: logits of each shard, the averaged gradients and the BatchNorm running
statistics.  Also: bench.py's self-launching strong-scaling + HIP-graph mode, rehearsed on two
ranks."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _make(kind):
    import connectome_gnn_amd as C
    cls, hidden = {"gcn64": (C.GCNConnectome, 64), "gcn128": (C.GCNConnectome, 128),
                   "sage64": (C.GraphSAGEConnectome, 64)}[kind]
    return cls(5, hidden, dropout=0.0)


def _worker(rank, world, port, q, kind="gcn64", backend="gloo"):
    sys.path.insert(0, ROOT)
    local = rank if backend == "nccl" else 0
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(local), HSA_ENABLE_IPC_MODE_LEGACY="0")
    try:
        import torch.distributed as dist
        import connectome_gnn_amd as C
        from connectome_gnn_amd import dist as cdist
        from connectome_gnn_amd.graph import shard_slice
        cdist.init_from_env(backend=backend)
        torch.cuda.set_device(local)
        graphs = C.generate_dataset(8, 84, 8, seed=21)
        torch.manual_seed(5)
        model = _make(kind).to("cuda").train()
        cdist.broadcast_parameters(model)
        model = cdist.convert_sync_batchnorm(model)
        sync = cdist.GradSync(model.parameters())
        mine = shard_slice(list(range(8)), rank, world)
        b = C.collate_graphs([graphs[i] for i in mine]).to("cuda")
        sync.zero_grad()
        logits = model(b)
        assert model.impl_used == "fused"
        torch.nn.functional.cross_entropy(logits, b.labels).backward()
        sync()
        # numpy: pickled by value (torch tensors would travel as shared-memory handles that die
        # with this process)
        out = {"logits": logits.detach().cpu().numpy(),
               "grads": {k: p.grad.detach().cpu().numpy().copy() for k, p in model.named_parameters()},
               "rm": model.batch_norms[2].running_mean.cpu().numpy().copy(),
               "rv": model.batch_norms[2].running_var.cpu().numpy().copy()}
        dist.barrier()
        q.put((rank, out))
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc()))
    finally:
        import torch.distributed as dist
        if dist.is_initialized():
            dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("backend", ["gloo", "nccl"])
@pytest.mark.parametrize("kind", ["gcn64", "gcn128", "sage64"])
def test_two_rank_sync_bn_equals_single_process_full_batch(kind, backend):
    """All three one-node encoders (per-tile GCN, wide GCN, GraphSAGE) under SyncBatchNorm."""
    if backend == "nccl" and torch.cuda.device_count() < 2:
        pytest.skip("RCCL needs one device per rank: fewer than 2 GPUs visible")
    import connectome_gnn_amd as C
    graphs = C.generate_dataset(8, 84, 8, seed=21)
    torch.manual_seed(5)
    ref = _make(kind).to("cuda").train()
    full = C.collate_graphs(graphs).to("cuda")
    lg = ref(full)
    torch.nn.functional.cross_entropy(lg, full.labels).backward()

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, kind, backend)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(60)
    for r in (0, 1):
        assert not isinstance(res[r], str), res[r]
    T = torch.from_numpy
    got = torch.cat([T(res[0]["logits"]), T(res[1]["logits"])])
    torch.testing.assert_close(got, lg.detach().cpu(), rtol=1e-5, atol=2e-6)
    for k, p in ref.named_parameters():
        for r in (0, 1):
            w = p.grad.cpu()
            torch.testing.assert_close(T(res[r]["grads"][k]), w, rtol=1e-4,
                                       atol=2e-6 + 1e-5 * float(w.abs().max()), msg=lambda s: f"{k}: {s}")
    torch.testing.assert_close(T(res[0]["rm"]), ref.batch_norms[2].running_mean.cpu(), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(T(res[0]["rv"]), ref.batch_norms[2].running_var.cpu(), rtol=1e-5, atol=1e-6)


@pytest.mark.timeout(600)
@pytest.mark.parametrize("mode", ["strong-graph-split", "strong-eager", "weak-graph-split"])
def test_bench_self_launch_two_ranks(mode):
    """`python bench.py --gpus 2` (no torchrun around it) starts its own two ranks, cuts the global
    batch in two (strong) or keeps it per rank (weak), replays the step as HIP graphs with the
    gradient all-reduce between them, and rank 0 prints the JSON line.  Rehearsal on one device
    over gloo (--one-device); on a multi-GPU node the same command runs over RCCL."""
    import json
    import subprocess
    scaling, launch = mode.split("-")[0], mode.split("-")[1]
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2",
           "--workload", "cfg2-gcn-512x84-h64", "--batch", "64", "--scaling", scaling, "--launch", launch,
           "--no-cpu-baseline", "--one-device", "--backend", "gloo"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=540, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == scaling
    assert out["config"]["global_batch"] == (64 if scaling == "strong" else 128)
    assert out["config"]["graphs_per_gpu"] == (32 if scaling == "strong" else 64)
    assert out["config"]["launch"].startswith("hip-graph" if launch == "graph" else "eager")
    assert out["value"] > 0 and out["final_loss"] == out["final_loss"]      # finite, not NaN


@pytest.mark.parametrize("kind", ["gcn64", "sage64", "gcn128"])
def test_direct_gradient_writes_equal_autograd_accumulation(kind):
    """dist.GradSync(direct=True): the one-node GCN encoder and the classifier + loss launch write their
    parameter gradients straight into the zeroed views of the flat all-reduce buffer (ops.grad_destination)
    instead of returning them to autograd, which would add each onto its view with a launch of its own.  Same
    kernels, another destination: bit-identical to direct=False and to plain `.grad` (no GradSync); a second
    micro-batch before the next zero_grad accumulates on top through autograd (gradient accumulation)."""
    import connectome_gnn_amd as C
    from connectome_gnn_amd import dist as cdist, ops
    b1 = C.collate_graphs(C.generate_dataset(12, 84, 8, seed=4)).to("cuda")
    b2 = C.collate_graphs(C.generate_dataset(12, 84, 8, seed=5)).to("cuda")
    loss_fn = ops.CrossEntropyLoss()
    got = {}
    for mode in ("plain", "views", "direct"):
        torch.manual_seed(1)
        m = _make(kind).to("cuda").train()
        sync = None if mode == "plain" else cdist.GradSync(m.parameters(), direct=(mode == "direct"))
        if sync is not None:
            sync.zero_grad()
        ops.backward_unit(ops.model_loss(m, loss_fn, b1))
        one = [p.grad.detach().clone() for p in m.parameters()]
        if mode == "direct":
            flat_ptr = sync.flat.data_ptr()
            assert all(flat_ptr <= p.grad.data_ptr() < flat_ptr + 4 * sync.numel for p in m.parameters())
            # every parameter's destination was claimed by a hand-written backward (encoder or classifier + loss)
            assert not any(getattr(p, "_cgnn_direct", False) for p in m.parameters())
        ops.backward_unit(ops.model_loss(m, loss_fn, b2))          # second micro-batch, no zero_grad in between
        got[mode] = (one, [p.grad.detach().clone() for p in m.parameters()])
    for mode in ("views", "direct"):
        for (a, c), name in zip(zip(got[mode][0], got["plain"][0]), [n_ for n_, _ in _make(kind).named_parameters()]):
            assert torch.equal(a, c), (mode, name)
        for a, c in zip(got[mode][1], got["plain"][1]):
            torch.testing.assert_close(a, c, rtol=1e-6, atol=1e-7)       # (a + b summed in another order)


def _trainer_worker(rank, world, port, q, resident):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    try:
        import torch.distributed as dist
        import connectome_gnn_amd as C
        from connectome_gnn_amd import dist as cdist
        cdist.init_from_env(backend="gloo")
        torch.cuda.set_device(0)
        graphs = C.generate_dataset(50, 84, 8, seed=13)
        torch.manual_seed(3)
        model = C.GCNConnectome(5, 64, dropout=0.0).to("cuda")
        cdist.broadcast_parameters(model)
        sync = cdist.GradSync(model.parameters())
        opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-4)
        kw = {} if resident else {"resident": False, "graph": False}
        tr = C.Trainer(model, opt, device="cuda", grad_sync=sync, **kw)
        ld = C.ConnectomeDataLoader(graphs[:40], batch_size=16, shuffle=True, rank=rank, world_size=world)
        vl = C.ConnectomeDataLoader(graphs[40:], batch_size=10, shuffle=False, rank=rank, world_size=world)
        hist = tr.fit(ld, vl, num_epochs=3, patience=5, verbose=False)
        out = {"hist": hist, "graph": bool(tr.graph), "graphs": len(tr._graphs),
               "w": {k: v.detach().cpu().numpy().copy() for k, v in model.state_dict().items()}}
        dist.barrier()
        q.put((rank, out))
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc()))
    finally:
        import torch.distributed as dist
        if dist.is_initialized():
            dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_trainer_defaults_over_the_reference_loader_equal_the_host_path():
    """Graph-sharded data parallelism through the drop-in API with its defaults: two ranks, each
    `Trainer(model, torch.optim.Adam(...), grad_sync=GradSync(...))` over the reference's list-backed loader with
    (rank, world_size) -- the dataset packed into HBM per rank, the shard's batches assembled on the device, one
    captured step per shard size with the gradient all-reduce between its two graphs, gradients written straight
    into the flat buffer -- against the same two ranks on the host-collate / eager path: same curves, same weights
    on both ranks (per-rank BatchNorm both ways).  gloo on one shared device; RCCL runs the same code."""
    ctx = mp.get_context("spawn")
    runs = {}
    for resident in (False, True):
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_trainer_worker, args=(r, 2, port, q, resident)) for r in range(2)]
        for p in procs:
            p.start()
        res = dict(q.get(timeout=300) for _ in procs)
        for p in procs:
            p.join(60)
        for r in (0, 1):
            assert not isinstance(res[r], str), res[r]
        runs[resident] = res
    T = torch.from_numpy
    for r in (0, 1):
        assert runs[True][r]["graph"] and runs[True][r]["graphs"] >= 2          # shards of 8 and 4 graphs (+ eval steps)
        assert not runs[False][r]["graph"]
        for key in ("train_loss", "val_loss", "val_acc"):
            torch.testing.assert_close(torch.tensor(runs[True][r]["hist"][key]), torch.tensor(runs[False][r]["hist"][key]),
                                       rtol=2e-4, atol=2e-6, msg=lambda s: f"rank {r} {key}: {s}")
    # both ranks hold the same weights (they applied the same averaged gradients), and the two paths agree
    for k, v in runs[True][0]["w"].items():
        if v.dtype.kind != "f":
            continue
        if "running" not in k:                    # parameters: identical on the two ranks (running statistics are per rank)
            assert np.array_equal(runs[True][1]["w"][k], v), f"{k} differs between the ranks"
        if not (k.startswith("convs.") and k.endswith(".bias")) and "running_mean" not in k:
            torch.testing.assert_close(T(v), T(runs[False][0]["w"][k]), rtol=2e-3, atol=2e-5, msg=lambda s: f"{k}: {s}")
