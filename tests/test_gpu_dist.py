"""N > 1 parity on the GPU box: two processes share the one GPU and talk over gloo (RCCL refuses
two ranks on one device; the collectives' semantics are the same).  With SyncBatchNorm the
graph-sharded run must reproduce the single-process full-batch run: logits of each shard, the
averaged gradients and the BatchNorm running statistics."""
import os
import socket
import sys

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


def _worker(rank, world, port, q, kind="gcn64"):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK="0")
    try:
        import torch.distributed as dist
        import connectome_gnn_amd as C
        from connectome_gnn_amd import dist as cdist
        from connectome_gnn_amd.graph import shard_slice
        cdist.init_from_env(backend="gloo")
        torch.cuda.set_device(0)
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
@pytest.mark.parametrize("kind", ["gcn64", "gcn128", "sage64"])
def test_two_rank_sync_bn_equals_single_process_full_batch(kind):
    """All three one-node encoders (per-tile GCN, wide GCN, GraphSAGE) under SyncBatchNorm."""
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
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, kind)) for r in range(2)]
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
