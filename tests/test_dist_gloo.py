"""world_size-2 tests of the data-parallel host path over gloo (CPU tensors): the flat-buffer
gradient all-reduce, parameter broadcast, sharded loaders and the sync-BN sum exchange used by
the fused encoder.  The HIP kernels are not involved (no GPU here); what is checked is that
mean-of-shard-gradients equals the global-batch gradient, using the oracle as the local model."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import connectome_gnn_amd as C
    from connectome_gnn_amd import dist as cdist
    from connectome_gnn_amd.fused import _sync_sums
    from connectome_gnn_amd.graph import shard_slice
    from oracle import reference_path as O
    try:
        r, w, _ = cdist.init_from_env(backend="gloo")
        assert (r, w) == (rank, world)
        torch.set_num_threads(1)
        graphs = C.generate_dataset(8, 20, 4, seed=3)
        # --- parameters: different per rank, then broadcast from rank 0
        torch.manual_seed(100 + rank)
        lin = torch.nn.Linear(5, 3)
        cdist.broadcast_parameters(lin)
        ref = [p.detach().clone() for p in lin.parameters()]
        gathered = [torch.zeros_like(ref[0]) for _ in range(world)]
        dist.all_gather(gathered, ref[0])
        assert all(torch.equal(g, gathered[0]) for g in gathered)
        # --- gradient sync: oracle model with local (per-shard) BN replaced by eval-mode BN so
        # that shard gradients average exactly to the global-batch gradient
        torch.manual_seed(7)
        st = O.require_grad(O.init_gcn_state(5, 16))
        params = [st[k] for k in O.param_keys(st)]
        sync = cdist.GradSync(params)

        def grads_on(gs):
            for p in params:
                p.grad = None
            b = O.collate([g.node_features for g in gs], [g.edge_index for g in gs],
                          [g.edge_weight for g in gs], [g.label for g in gs])
            loss = torch.nn.functional.cross_entropy(O.gcn_forward(st, b, 0.0, False), b.labels)
            loss.backward()
            return [p.grad.clone() for p in params]

        full = grads_on(graphs)
        mine = shard_slice(list(range(8)), rank, world)
        grads_on([graphs[i] for i in mine])
        sync()                                                  # equal shards: plain mean
        for p, g in zip(params, full):
            torch.testing.assert_close(p.grad, g, rtol=1e-5, atol=1e-7)
        # unequal shards (5 + 3): weight by share of the global batch
        cut = [graphs[:5], graphs[5:]][rank]
        grads_on(cut)
        sync(local_graphs=len(cut), global_graphs=8)
        for p, g in zip(params, full):
            torch.testing.assert_close(p.grad, g, rtol=1e-5, atol=1e-7)
        # --- sync-BN sum exchange of the fused encoder: [sum | sumsq] + row count
        buf = torch.cat([torch.arange(128, dtype=torch.float64) * (rank + 1),
                         torch.tensor([10.0 * (rank + 1)], dtype=torch.float64)])
        _sync_sums(buf, None)
        assert float(buf[128]) == 30.0 and torch.equal(buf[:128], torch.arange(128, dtype=torch.float64) * 3)
        # --- grads as permanent views of the flat buffer: zero_grad + in-place accumulation
        sync.zero_grad()
        b = O.collate([g.node_features for g in [graphs[i] for i in mine]],
                      [g.edge_index for g in [graphs[i] for i in mine]],
                      [g.edge_weight for g in [graphs[i] for i in mine]],
                      [g.label for g in [graphs[i] for i in mine]])
        torch.nn.functional.cross_entropy(O.gcn_forward(st, b, 0.0, False), b.labels).backward()
        lo, hi = sync.flat.data_ptr(), sync.flat.data_ptr() + sync.flat.numel() * 4
        assert all(lo <= p.grad.data_ptr() < hi for p in params)      # still views: no copies
        sync()
        for p, g in zip(params, full):
            torch.testing.assert_close(p.grad, g, rtol=1e-5, atol=1e-7)
        # --- sharded loader: same global order on all ranks, disjoint contiguous shards
        torch.manual_seed(11)
        ld = C.ConnectomeDataLoader(graphs, batch_size=4, shuffle=True, rank=rank, world_size=world)
        first = [b.node_features.clone() for b in ld]
        sizes = torch.tensor([b.shape[0] for b in first])
        allsz = [torch.zeros_like(sizes) for _ in range(world)]
        dist.all_gather(allsz, sizes)
        assert int(sum(a.sum() for a in allsz)) == 8 * 20
        dist.barrier()
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc()))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_data_parallel_host_path_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=150) for _ in procs]
    for p in procs:
        p.join(30)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


class _PoolLinear(torch.nn.Module):
    """CPU stand-in for a connectome model (the package has no CPU message-passing path): mean of
    the node features per graph -> Linear.  Enough to drive Trainer's data-parallel host logic."""

    def __init__(self):
        super().__init__()
        self.lin = torch.nn.Linear(5, 2)

    def forward(self, batch):
        sums = torch.zeros(batch.num_graphs, 5).index_add_(0, batch.batch, batch.node_features)
        return self.lin(sums / (batch.ptr[1:] - batch.ptr[:-1]).clamp(min=1).unsqueeze(1))


def _trainer_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import connectome_gnn_amd as C
    from connectome_gnn_amd import dist as cdist
    try:
        cdist.init_from_env(backend="gloo")
        torch.set_num_threads(1)
        graphs = C.generate_dataset(23, 12, 4, seed=5)
        train, val = graphs[:15], graphs[15:]

        def run(r, w, sync_cls):
            torch.manual_seed(1)
            model = _PoolLinear()
            opt = torch.optim.SGD(model.parameters(), lr=0.5)
            sync = sync_cls(model.parameters()) if sync_cls else None
            tr = C.Trainer(model, opt, device="cpu", grad_sync=sync, loss_fn=torch.nn.CrossEntropyLoss())
            # batch_size 5 over 2 ranks: shards of 3 + 2 graphs in every batch; val 8 graphs -> 3 + 2, 2 + 1
            hist = tr.fit(C.ConnectomeDataLoader(train, batch_size=5, shuffle=False, rank=r, world_size=w),
                          C.ConnectomeDataLoader(val, batch_size=5, shuffle=False, rank=r, world_size=w),
                          num_epochs=40, patience=2, verbose=False)
            return hist, [p.detach().clone() for p in model.parameters()], tr.evaluate(
                C.ConnectomeDataLoader(val, batch_size=5, shuffle=False, rank=r, world_size=w))

        hist, params, ev = run(rank, world, cdist.GradSync)
        # every rank saw the same curves, stopped at the same epoch and holds the same weights
        mine = torch.tensor(hist["val_loss"] + [float(len(hist["val_loss"]))], dtype=torch.float64)
        both = [torch.zeros_like(mine) for _ in range(world)]
        # lengths are equal iff the ranks stopped together; all_gather would hang/raise otherwise
        dist.all_gather(both, mine)
        assert torch.equal(both[0], both[1])
        for p in params:
            g = [torch.zeros_like(p) for _ in range(world)]
            dist.all_gather(g, p)
            assert torch.equal(g[0], g[1])
        assert ev["total"] == 8                               # global count, not this rank's shard
        # ... and they are the single-process run on the unsharded batches: unequal shards are
        # weighted by their graph counts, so the update is the global-batch gradient
        torch.distributed.barrier()
        ref_hist, ref_params, ref_ev = run(0, 1, None)
        assert len(ref_hist["val_loss"]) == len(hist["val_loss"]) < 40      # early stop happened
        torch.testing.assert_close(torch.tensor(hist["train_loss"]), torch.tensor(ref_hist["train_loss"]),
                                   rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(torch.tensor(hist["val_loss"]), torch.tensor(ref_hist["val_loss"]),
                                   rtol=1e-5, atol=1e-6)
        for a, b_ in zip(params, ref_params):
            torch.testing.assert_close(a, b_, rtol=1e-4, atol=1e-6)
        assert ev["correct"] == ref_ev["correct"]
        # path agreement helper: False anywhere -> False everywhere
        assert cdist.agree(True, "cpu") is True
        assert cdist.agree(rank == 0, "cpu") is False
        q.put((rank, "ok"))
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc()))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_trainer_data_parallel_unequal_shards_and_early_stop_world2():
    """Trainer weights gradients by graph count (partial/odd batches), reduces its epoch
    tallies over ranks before reading them, so every rank early-stops at the same epoch, restores
    the same weights, and the trajectory equals the single-process run."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_trainer_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=150) for _ in procs]
    for p in procs:
        p.join(30)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


def _worker8(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from connectome_gnn_amd import dist as cdist
    from connectome_gnn_amd.graph import shard_slice
    try:
        r, w, _ = cdist.init_from_env(backend="gloo")
        assert (r, w) == (rank, world) and dist.get_world_size() == world
        torch.set_num_threads(1)
        params = [torch.zeros(37, 5, requires_grad=True), torch.zeros(11, requires_grad=True)]
        sync = cdist.GradSync(params)
        assert sync.world == world and not sync._avg          # gloo: scale + SUM
        for total in (4096, 4100):                            # bench.py --scaling strong at N = 8
            shards = [shard_slice(list(range(total)), rr, world) for rr in range(world)]
            sizes = [len(s_) for s_ in shards]
            assert sum(sizes) == total and max(sizes) - min(sizes) <= 1
            assert [i for s_ in shards for i in s_] == list(range(total))      # contiguous, in rank order
            equal = len(set(sizes)) == 1
            assert equal == (total % world == 0)
            sync.zero_grad()
            for pi, p in enumerate(params):                   # rank-dependent "gradient", written in place
                p.grad.add_(torch.full_like(p, float(rank + 1)) * (pi + 1))
            sync(local_graphs=None if equal else sizes[rank])
            want = sum(sizes[rr] * (rr + 1) for rr in range(world)) / total
            for pi, p in enumerate(params):
                torch.testing.assert_close(p.grad, torch.full_like(p, want * (pi + 1)), rtol=1e-6, atol=0)
            lo, hi = sync.flat.data_ptr(), sync.flat.data_ptr() + sync.flat.numel() * 4
            assert all(lo <= p.grad.data_ptr() < hi for p in params)
        assert cdist.agree(True, "cpu") is True and cdist.agree(rank != 5, "cpu") is False
        assert cdist.agree_min(rank + 3, "cpu") == 3
        dist.barrier()
        q.put((rank, "ok"))
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc()))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_grad_sync_and_sharding_world8_rehearsal():
    """The 8-rank layout bench.py --gpus 8 runs (nobody can rehearse it on hardware): shard_slice of a
    4096- and a 4100-graph global batch, GradSync with equal and unequal shards, the agreement
    helpers that decide graph-vs-eager together -- eight gloo ranks on CPU tensors."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker8, args=(r, 8, port, q)) for r in range(8)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(30)
    assert sorted(res) == [(r, "ok") for r in range(8)], res
