"""Config 1 of BASELINE.json: the reference's demo experiment run on the package
(examples/demo.py -- same data recipe, split, models, optimiser and schedule as
/root/reference/examples/demo.py:35-134).  The reference documents 55-70 % test accuracy
(README.md:115) on its own random stream; ours differs in the stream only (own generator, GPU
dropout), so the check is the neighbourhood of that band plus the exact parameter counts."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_demo_experiment_trains_both_models():
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    import demo
    res = demo.run(device="cuda", epochs=30, verbose=False)
    gcn, sage = res["GCNConnectome"], res["GraphSAGEConnectome"]
    assert gcn["params"] == 11_234 and sage["params"] == 19_746
    assert gcn["impl"] == "fused" and sage["impl"] == "fused"        # the HIP encoders ran
    for r in (gcn, sage):
        t = r["test"]
        assert t["total"] == 45
        # 45 test subjects: one subject is 2.2 points; the reference's band is 55-70 %
        assert 0.45 <= t["accuracy"] <= 0.80, t
        h = r["history"]
        assert len(h["train_loss"]) == len(h["val_loss"]) == len(h["val_acc"]) >= 8
        assert all(torch.isfinite(torch.tensor(h["train_loss"])))
        assert min(h["train_loss"]) < h["train_loss"][0]            # it learned something


def test_trainer_graph_mode_matches_eager_on_cached_batches():
    """Trainer(graph=True) over a ResidentDataLoader with cached batches: each batch's step is
    captured once and replayed; with dropout 0 the trajectory equals the eager Trainer's."""
    import connectome_gnn_amd as C
    from connectome_gnn_amd.resident import ResidentDataLoader
    from connectome_gnn_amd.synthetic import generate_packed
    ds = generate_packed(96, 84, 8, seed=3).to("cuda")
    hist = {}
    for mode in ("eager", "graph"):
        torch.manual_seed(0)
        m = C.GCNConnectome(5, 64, dropout=0.0)
        opt = torch.optim.Adam(m.parameters(), lr=1e-3, weight_decay=1e-4, capturable=(mode == "graph"))
        tr = C.Trainer(m, opt, device="cuda", graph=(mode == "graph"))
        ld = ResidentDataLoader(ds, batch_size=32, shuffle=False, cache_batches=True, prepare=tr.model.prepare_batch)
        hist[mode] = [tr.train_epoch(ld) for _ in range(4)]
        if mode == "graph":
            assert len(tr._graphs) == 3                              # one captured graph per batch
        ev = tr.evaluate(ld)
        assert ev["total"] == 96
    torch.testing.assert_close(torch.tensor(hist["graph"]), torch.tensor(hist["eager"]), rtol=2e-5, atol=1e-6)
    # batches that never come round again are not captured (nothing accumulates), and the number of
    # kept steps is bounded: beyond max_graphs a repeating batch stays eager
    m = C.GCNConnectome(5, 64, dropout=0.0)
    opt = torch.optim.Adam(m.parameters(), lr=1e-3, capturable=True)
    tr = C.Trainer(m, opt, device="cuda", graph=True, max_graphs=2)
    fresh = ResidentDataLoader(ds, batch_size=32, shuffle=True, prepare=tr.model.prepare_batch)
    for _ in range(3):
        tr.train_epoch(fresh)
    assert len(tr._graphs) == 0 and len(tr._seen) <= 16
    ld = ResidentDataLoader(ds, batch_size=32, shuffle=False, cache_batches=True, prepare=tr.model.prepare_batch)
    for _ in range(3):
        tr.train_epoch(ld)
    assert len(tr._graphs) == 2
    tr.clear_graphs()
    assert not tr._graphs
    with pytest.raises(ValueError):
        C.Trainer(C.GCNConnectome(5, 64), torch.optim.Adam(C.GCNConnectome(5, 64).parameters()), graph=True)


def test_subject_structure_cache_equals_per_batch_build():
    """ResidentDataLoader(structure_cache=True): a batch assembled from the per-subject cache
    (blocked-ELL block offsets + `dis` gathered, entries never copied, COO fields lazy) trains
    bit-identically to the same subjects through assemble_batch + the per-batch builders."""
    import connectome_gnn_amd as C
    from connectome_gnn_amd.resident import ResidentDataLoader, assemble_batch
    from connectome_gnn_amd.structure_cache import ResidentBatch
    from connectome_gnn_amd.synthetic import generate_packed
    ds = generate_packed(24, 360, 14, seed=2).to("cuda")
    torch.manual_seed(3)
    ld = ResidentDataLoader(ds, batch_size=8, shuffle=True, structure_cache=True)
    batches = list(ld)
    assert len(batches) == 3 and all(isinstance(b, ResidentBatch) for b in batches)
    outs = []
    for use_cache in (True, False):
        torch.manual_seed(0)
        m = C.GCNConnectome(5, 64, dropout=0.3).to("cuda").train()
        opt = torch.optim.Adam(m.parameters(), lr=1e-3)
        res = []
        for rb in batches:
            b = rb if use_cache else assemble_batch(ds, rb._ids)
            opt.zero_grad()
            lg = m(b)
            assert m.impl_used == "fused"
            torch.nn.functional.cross_entropy(lg, b.labels).backward()
            opt.step()
            res.append(lg.detach().clone())
        outs.append((res, [p.detach().clone() for p in m.parameters()]))
    for a, c in zip(outs[0][0], outs[1][0]):
        assert torch.equal(a, c)
    for a, c in zip(outs[0][1], outs[1][1]):
        assert torch.equal(a, c)
    # prefetched on the side stream (as Trainer.train_epoch consumes it): the same batches
    torch.manual_seed(3)
    ld2 = ResidentDataLoader(ds, batch_size=8, shuffle=True, structure_cache=True, prefetch=True)
    for rb, rb2 in zip(batches, ld2):
        assert torch.equal(rb._ids, rb2._ids) and torch.equal(rb.node_features, rb2.node_features)
        assert torch.equal(rb.structure().gcn_dis(None), rb2.structure().gcn_dis(None))
    # the lazily assembled COO is the ordinary one
    ref = assemble_batch(ds, batches[0]._ids)
    assert batches[0]._coo is None
    assert torch.equal(batches[0].edge_index, ref.edge_index) and torch.equal(batches[0].batch, ref.batch)
    # paths the cache has no arrays for (it carries no CSR) are refused loudly, not served a wrong structure
    with pytest.raises((AttributeError, ValueError, RuntimeError)):
        C.GCNConnectome(5, 128).to("cuda")(batches[1])
    with pytest.raises(ValueError):       # a graph must fit one LDS tile
        ResidentDataLoader(generate_packed(2, 400, 8, seed=1).to("cuda"), batch_size=2, structure_cache=True)


@pytest.mark.parametrize("n,k", [(84, 8), (48, 6), (192, 10)])
def test_subject_structure_cache_small_graphs_get_a_tile_each(n, k):
    """Graphs of <= 192 nodes share tiles in the per-batch build (their 16-row blocks straddle graphs);
    under the subject cache each gets its own tile.  Same mathematics, another grouping of the
    BatchNorm / weight-gradient partial sums: equal to the per-batch build at rounding level."""
    import connectome_gnn_amd as C
    from connectome_gnn_amd.resident import ResidentDataLoader, assemble_batch
    from connectome_gnn_amd.synthetic import generate_packed
    ds = generate_packed(40, n, k, seed=5).to("cuda")
    torch.manual_seed(3)
    batches = list(ResidentDataLoader(ds, batch_size=16, shuffle=True, structure_cache=True))
    assert [b.num_graphs for b in batches] == [16, 16, 8]
    st = batches[0].structure()
    assert int(st.fused_meta(384, 256).tile_ptr.numel()) - 1 == 16          # one tile per graph
    outs = []
    for use_cache in (True, False):
        torch.manual_seed(0)
        m = C.GCNConnectome(5, 64, dropout=0.0).to("cuda").train()
        opt = torch.optim.Adam(m.parameters(), lr=1e-3)
        res = []
        for rb in batches:
            b = rb if use_cache else assemble_batch(ds, rb._ids)
            opt.zero_grad()
            lg = m(b)
            assert m.impl_used == "fused"
            torch.nn.functional.cross_entropy(lg, b.labels).backward()
            res.append((lg.detach().clone(), [p.grad.detach().clone() for p in m.parameters()]))
            opt.step()
        outs.append(res)
    names = [k_ for k_, _ in C.GCNConnectome(5, 64).named_parameters()]
    for (la, ga), (lb, gb) in zip(*outs):
        torch.testing.assert_close(la, lb, rtol=1e-5, atol=2e-6)
        for nm, a, c in zip(names, ga, gb):
            if nm.startswith("convs.") and nm.endswith(".bias"):
                continue                      # exactly-zero true gradient: rounding noise either way
            scale = float(c.abs().max())
            assert float((a - c).abs().max()) <= 2e-5 * scale + 2e-6, nm


@pytest.mark.parametrize("n,k,hidden", [(360, 14, 128), (84, 8, 64)])
def test_subject_structure_cache_serves_graphsage(n, k, hidden):
    """The cache's GraphSAGE family (blocked-ELL without self-loops, `den` per subject; built on first
    use) and the layer-0 mean through the tiled aggregate on a padded panel (the cache carries no
    CSR): the same training steps as through assemble_batch + the per-batch builders."""
    import connectome_gnn_amd as C
    from connectome_gnn_amd.resident import ResidentDataLoader, assemble_batch
    from connectome_gnn_amd.synthetic import generate_packed
    ds = generate_packed(24, n, k, seed=7).to("cuda")
    torch.manual_seed(3)
    batches = list(ResidentDataLoader(ds, batch_size=8, shuffle=True, structure_cache=True))
    outs = []
    for use_cache in (True, False):
        torch.manual_seed(0)
        m = C.GraphSAGEConnectome(5, hidden, dropout=0.0).to("cuda").train()
        opt = torch.optim.Adam(m.parameters(), lr=1e-3)
        res = []
        for rb in batches:
            b = rb if use_cache else assemble_batch(ds, rb._ids)
            opt.zero_grad()
            lg = m(b)
            assert m.impl_used == "fused"
            torch.nn.functional.cross_entropy(lg, b.labels).backward()
            res.append((lg.detach().clone(), [p.grad.detach().clone() for p in m.parameters()]))
            opt.step()
        outs.append(res)
    assert set(batches[0]._cache._fam) == {"gcn", "sage"}
    for (la, ga), (lb, gb) in zip(*outs):
        torch.testing.assert_close(la, lb, rtol=2e-5, atol=5e-6)
        for a, c in zip(ga, gb):
            assert float((a - c).abs().max()) <= 5e-5 * float(c.abs().max()) + 5e-6


def test_trainer_graph_mode_replays_fresh_shuffled_resident_batches():
    """Trainer(graph=True) over a ResidentDataLoader(shuffle=True, structure_cache=True): every epoch
    re-draws the batch compositions (the reference's loader semantics, graph.py:190-197), yet all
    batches of one size share ONE captured step -- the batch is assembled inside the graph from its
    subject ids (graphed.GraphedResidentStep) -- and the trajectory equals the eager Trainer's."""
    import connectome_gnn_amd as C
    from connectome_gnn_amd.graphed import GraphedResidentStep
    from connectome_gnn_amd.resident import ResidentDataLoader
    from connectome_gnn_amd.synthetic import generate_packed
    _trainer_replay_equals_eager(generate_packed(40, 360, 14, seed=4).to("cuda"))
    # 84-ROI graphs (BASELINE config 2): one tile per graph under the cache, same replay
    _trainer_replay_equals_eager(generate_packed(40, 84, 8, seed=4).to("cuda"), masks=False)
    # GraphSAGE over the same kind of loader (the cache's second family)
    _trainer_replay_equals_eager(generate_packed(40, 84, 8, seed=4).to("cuda"), masks=False, cls="sage")
    # long epochs of a small step: runs of four consecutive steps replay as ONE graph (GraphedTrainStep.capture_run);
    # 168 subjects = ten batches of 16 (two run-graphs + two single steps) and one of 8
    runs = _trainer_replay_equals_eager(generate_packed(168, 84, 8, seed=6).to("cuda"), masks=False)
    assert runs == [True, False]                 # the 16-graph step got its run-graph, the 8-graph tail step did not


def _trainer_replay_equals_eager(ds, masks=True, cls="gcn"):
    import connectome_gnn_amd as C
    from connectome_gnn_amd.graphed import GraphedResidentStep
    from connectome_gnn_amd.resident import ResidentDataLoader
    hist, finals = {}, {}
    for mode in ("eager", "graph"):
        torch.manual_seed(0)
        m = (C.GCNConnectome if cls == "gcn" else C.GraphSAGEConnectome)(5, 64, dropout=0.0)
        opt = torch.optim.Adam(m.parameters(), lr=1e-3, weight_decay=1e-4, capturable=True)   # the same update both ways
        tr = C.Trainer(m, opt, device="cuda", graph=(mode == "graph"))
        ld = ResidentDataLoader(ds, batch_size=16, shuffle=True, structure_cache=True, prefetch=True,
                                prepare=tr.model.prepare_batch)
        hist[mode] = [tr.train_epoch(ld) for _ in range(4)]          # batches of 16, 16, 8 subjects
        finals[mode] = {k: v.detach().clone() for k, v in tr.model.state_dict().items()}
        if mode == "graph":
            assert len(tr._graphs) == 2                               # one per batch SIZE, not per batch
            assert all(isinstance(s, GraphedResidentStep) for s in tr._graphs.values())
            run_graphs = [s._run_graph is not None for _, s in sorted(tr._graphs.items(), key=lambda kv: -kv[0][2])]
    torch.testing.assert_close(torch.tensor(hist["graph"]), torch.tensor(hist["eager"]), rtol=2e-5, atol=1e-6)
    for k, v in finals["eager"].items():
        # (GCN's bias ahead of BatchNorm has a zero true gradient -- Adam moves it by +-lr per step on the SIGN
        # of rounding noise -- and the layer's running mean contains it; the replayed step uses one set of
        # layer-0 centring constants per dataset, the eager one per batch: equal functions, different rounding)
        if cls == "gcn" and (k.startswith("convs.") and k.endswith(".bias") or "running_mean" in k):
            continue
        if v.is_floating_point() and not (k.startswith("convs.") and k.endswith(".bias")):
            torch.testing.assert_close(finals["graph"][k], v, rtol=1e-4, atol=1e-5, msg=lambda s: f"{k}: {s}")
    if not masks:
        return run_graphs
    # with dropout the replays draw fresh masks: two epochs over the same subjects differ
    torch.manual_seed(1)
    m = C.GCNConnectome(5, 64, dropout=0.5)
    opt = torch.optim.Adam(m.parameters(), lr=0.0, capturable=True)      # frozen weights: only masks change
    tr = C.Trainer(m, opt, device="cuda", graph=True)
    ld = ResidentDataLoader(ds, batch_size=40, shuffle=False, structure_cache=True, prepare=tr.model.prepare_batch)
    vals = {round(tr.train_epoch(ld), 7) for _ in range(4)}
    assert len(vals) >= 3, vals


@pytest.mark.parametrize("kind,hidden", [("gcn", 64), ("sage", 64), ("gcn", 32)])
def test_the_unchanged_reference_script_is_served_from_the_device(kind, hidden):
    """`Trainer(model, torch.optim.Adam(...), device)` over a list-backed
    `ConnectomeDataLoader` (reference examples/demo.py:92-134, graph.py:174-197) packs the dataset into HBM
    once, draws each epoch's permutation with the loader's own global-RNG call, assembles batches on the
    device and -- where the per-subject structure cache serves the encoder -- replays one captured step per
    batch size.  Same trajectory as the host loader (dropout 0: the masks' stream is the only thing that
    may differ), same RNG consumption; hidden 32 (no cache, layered path) is served eagerly."""
    import connectome_gnn_amd as C
    graphs = C.generate_dataset(72, 84, 8, seed=5)
    hist, rng_after, evals, trainers = {}, {}, {}, {}
    for mode in ("host", "default"):
        torch.manual_seed(3)
        cls = C.GCNConnectome if kind == "gcn" else C.GraphSAGEConnectome
        m = cls(5, hidden, dropout=0.0)
        opt = torch.optim.Adam(m.parameters(), lr=1e-3, weight_decay=1e-4)
        tr = C.Trainer(m, opt, device="cuda", **({"resident": False, "graph": False} if mode == "host" else {}))
        ld = C.ConnectomeDataLoader(graphs[:56], batch_size=16, shuffle=True)        # 3 x 16 + 8: two batch sizes
        vl = C.ConnectomeDataLoader(graphs[56:], batch_size=16, shuffle=False)
        hist[mode] = tr.fit(ld, vl, num_epochs=4, patience=10, verbose=False)
        evals[mode] = tr.evaluate(vl)
        rng_after[mode] = torch.get_rng_state()
        trainers[mode] = tr
    for key in ("train_loss", "val_loss", "val_acc"):
        torch.testing.assert_close(torch.tensor(hist["default"][key]), torch.tensor(hist["host"][key]),
                                   rtol=2e-4, atol=2e-6, msg=lambda s: f"{key}: {s}")
    assert evals["default"]["total"] == evals["host"]["total"] == 16
    assert evals["default"]["correct"] == evals["host"]["correct"]
    assert abs(evals["default"]["loss"] - evals["host"]["loss"]) <= 2e-4 * abs(evals["host"]["loss"]) + 2e-6
    assert torch.equal(rng_after["default"], rng_after["host"])      # the same randperm calls, nothing else drawn
    tr = trainers["default"]
    assert len(tr._resident) == 2 and all(v[2] is not None for v in tr._resident.values())
    served = hidden == 64
    assert tr.graph is served
    if served:
        assert sorted(k[2] for k in tr._graphs if k[0] == "resident") == [8, 16]     # one captured step per batch size
        assert sorted(k[2] for k in tr._graphs if k[0] == "eval") == [16]            # ... and the evaluation passes replay too
        assert all(g["capturable"] for g in tr.optimizer.param_groups)
    else:
        assert not tr._graphs
    assert not trainers["host"]._graphs and not trainers["host"]._resident


def test_irregular_datasets_keep_the_host_loader():
    import connectome_gnn_amd as C
    graphs = C.generate_dataset(8, 20, 4, seed=1) + C.generate_dataset(8, 35, 4, seed=2)
    m = C.GCNConnectome(5, 64, dropout=0.0)
    tr = C.Trainer(m, torch.optim.Adam(m.parameters(), lr=1e-3), device="cuda")
    ld = C.ConnectomeDataLoader(graphs, batch_size=8, shuffle=True)
    loss = tr.train_epoch(ld)
    assert loss == loss and list(tr._resident.values())[0][2] is None and not tr.graph


def test_out_of_range_label_raises_when_the_dataset_is_packed():
    """torch's cross-entropy raises on a target outside [0, C) (reference train.py:49); the packed path checks
    the labels once on the host instead of producing a NaN loss on the device."""
    import connectome_gnn_amd as C
    graphs = C.generate_dataset(8, 20, 4, seed=1)
    graphs[3] = C.ConnectomeGraph(graphs[3].node_features, graphs[3].edge_index, graphs[3].edge_weight,
                                  torch.tensor(2, dtype=torch.long))
    m = C.GCNConnectome(5, 64)
    tr = C.Trainer(m, torch.optim.Adam(m.parameters(), lr=1e-3), device="cuda")
    with pytest.raises(IndexError, match="out of bounds"):
        tr.train_epoch(C.ConnectomeDataLoader(graphs, batch_size=8, shuffle=False))


def test_plain_loader_path_with_twelve_features_and_one_loader_for_both_roles():
    """The packed path with 9..16 input features (the per-tile kernels' non-factored layer 0 over the subject
    cache) and the INTEGRATION.md idiom `fit(loader, loader)`: same trajectory as the host loader."""
    import connectome_gnn_amd as C
    base = C.generate_dataset(40, 84, 8, seed=9)
    g = torch.Generator().manual_seed(4)
    graphs = [C.ConnectomeGraph(torch.randn(gr.num_nodes, 12, generator=g), gr.edge_index, gr.edge_weight, gr.label)
              for gr in base]
    hist = {}
    for mode in ("host", "default"):
        torch.manual_seed(2)
        m = C.GCNConnectome(12, 64, dropout=0.0)
        tr = C.Trainer(m, torch.optim.Adam(m.parameters(), lr=1e-3), device="cuda",
                       **({"resident": False, "graph": False} if mode == "host" else {}))
        ld = C.ConnectomeDataLoader(graphs, batch_size=16, shuffle=True)
        hist[mode] = tr.fit(ld, ld, num_epochs=3, patience=5, verbose=False)
        if mode == "default":
            assert tr.graph and m.impl_used == "fused"
    for key in ("train_loss", "val_loss", "val_acc"):
        torch.testing.assert_close(torch.tensor(hist["default"][key]), torch.tensor(hist["host"][key]), rtol=2e-4, atol=2e-6)


def test_evaluation_replay_with_run_graphs_equals_eager():
    """Trainer.evaluate over a resident loader: captured evaluation steps, long passes through graphs of four
    consecutive steps (GraphedEvalStep._capture_run), against the eager evaluation of the same model."""
    import connectome_gnn_amd as C
    from connectome_gnn_amd.optim import Adam
    from connectome_gnn_amd.resident import ResidentDataLoader
    from connectome_gnn_amd.synthetic import generate_packed
    ds = generate_packed(150, 84, 8, seed=8).to("cuda")            # nine batches of 16 + one of 6
    torch.manual_seed(0)
    m = C.GCNConnectome(5, 64).to("cuda")
    res = {}
    for graph in (False, True):
        tr = C.Trainer(m, Adam(m.parameters(), lr=1e-3), device="cuda", graph=graph)
        ld = ResidentDataLoader(ds, batch_size=16, shuffle=False, structure_cache=True)
        res[graph] = [tr.evaluate(ld) for _ in range(2)][-1]
        if graph:
            steps = {k[2]: s for k, s in tr._graphs.items() if k[0] == "eval"}
            assert sorted(steps) == [6, 16] and steps[16]._run_graph is not None and steps[6]._run_graph is None
    assert res[True]["total"] == res[False]["total"] == 150 and res[True]["correct"] == res[False]["correct"]
    assert abs(res[True]["loss"] - res[False]["loss"]) <= 1e-5 * abs(res[False]["loss"]) + 1e-6
