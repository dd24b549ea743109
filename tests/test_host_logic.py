"""CPU tests: drop-in data API (mirrors the reference's tests/test_graph.py and
tests/test_synthetic.py assertions), the C-ABI symbol table, and the no-fallback rule."""
import ctypes
import math
import os
import re

import numpy as np
import pytest
import torch

import connectome_gnn_amd as C
from connectome_gnn_amd import _lib
from connectome_gnn_amd.graph import shard_slice
from connectome_gnn_amd.synthetic import NUM_REGIONS, generate_packed, small_world_stats
from tests import golden_util as G

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _simple_graph(n=10, e=20, f=4, label=0, seed=0):
    g = torch.Generator().manual_seed(seed)
    s, d = torch.randint(0, n, (e,), generator=g), torch.randint(0, n, (e,), generator=g)
    w = torch.rand(e, generator=g)
    return C.ConnectomeGraph(torch.randn(n, f, generator=g), torch.stack([torch.cat([s, d]), torch.cat([d, s])]),
                             torch.cat([w, w]), torch.tensor(label))


# ------------------------------------------------------------------------------- C ABI
def test_public_names_match_reference():
    assert C.__all__ == ["ConnectomeGraph", "ConnectomeBatch", "ConnectomeDataLoader",
                         "collate_graphs", "generate_connectome", "generate_dataset",
                         "REGION_NAMES", "GCNConnectome", "GraphSAGEConnectome", "Trainer"]


def test_library_exports_every_declared_symbol():
    """Every function declared in include/cgnn.h is exported by libcgnn_hip.so and bound in
    _lib.PROTOTYPES (no compute calls: this box has no GPU)."""
    hdr = open(os.path.join(ROOT, "include", "cgnn.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(cgnn_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 30
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in cgnn.h but not exported"
    assert declared == set(_lib.PROTOTYPES), declared ^ set(_lib.PROTOTYPES)
    loaded = _lib.load()
    assert loaded.cgnn_abi_version() == _lib.ABI_VERSION
    assert loaded.cgnn_build_target() == b"gfx950"
    assert loaded.cgnn_csr_workspace_bytes(1000, 5000) > 0
    assert loaded.cgnn_csr_workspace_bytes(-1, 0) < 0          # CGNN_EINVAL, no crash


def test_tiles_struct_layout_matches_header():
    t = _lib.CgnnTiles()
    assert ctypes.sizeof(t) == 8 + 4 + 4 + 7 * 8
    assert [f[0] for f in t._fields_] == ["num_nodes", "num_tiles", "max_tile_rows", "tile_ptr",
                                          "tile_blk", "blk_off_dst", "ent_dst", "blk_off_src",
                                          "ent_src", "dis"]


def test_no_cpu_fallback_and_no_oracle_import_in_product():
    """The product path must raise on CPU tensors and must never import oracle/."""
    b = C.collate_graphs(C.generate_dataset(2, 20, 4))
    for cls in (C.GCNConnectome, C.GraphSAGEConnectome):
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            cls(5, 16)(b)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        b.structure()
    pkg = os.path.join(ROOT, "connectome_gnn_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "import oracle" not in src and "from oracle" not in src, fn


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libcgnn_hip.so")
    with pytest.raises(_lib.CgnnError, match="no CPU/eager fallback"):
        _lib.load()


# ------------------------------------------------------------------- containers / collate
def test_graph_properties_and_helpers():
    g = _simple_graph()
    assert (g.num_nodes, g.num_edges, g.num_features) == (10, 40, 4)
    a = g.adjacency_matrix()
    assert a.shape == (10, 10) and torch.allclose(a, a.t())
    assert (g.degree() >= 0).all()
    assert g.to("cpu").subject_id == g.subject_id


def test_collate_shapes_offsets_ptr():
    g1, g2 = _simple_graph(10, 20, label=0), _simple_graph(7, 9, label=1, seed=1)
    b = C.collate_graphs([g1, g2])
    assert b.num_graphs == 2 and b.num_nodes == 17
    assert b.edge_index.shape == (2, 58) and b.edge_index.dtype == torch.int64
    assert b.batch.tolist() == [0] * 10 + [1] * 7
    assert b.ptr.tolist() == [0, 10, 17]
    assert int(b.edge_index[:, 40:].min()) >= 10          # second graph offset by N1
    assert b.labels.tolist() == [0, 1]
    unl = C.collate_graphs([C.ConnectomeGraph(g1.node_features, g1.edge_index, g1.edge_weight)])
    assert unl.labels is None


def test_collate_bit_exact_vs_golden_g1():
    d = G.load("g1_collate_8x20.npz")
    bd = G.group(d, "batch")
    gs = [C.ConnectomeGraph(*g) for g in G.split_graphs(bd, d["edge_counts"])]
    b = C.collate_graphs(gs)
    for f in ("node_features", "edge_index", "edge_weight", "batch", "labels", "ptr"):
        assert torch.equal(getattr(b, f), bd[f]), f


def test_loader_len_partial_batch_and_shuffle_rng():
    ds = [_simple_graph(seed=i) for i in range(10)]
    ld = C.ConnectomeDataLoader(ds, batch_size=4, shuffle=False)
    assert len(ld) == math.ceil(10 / 4)
    sizes = [b.num_graphs for b in ld]
    assert sizes == [4, 4, 2]
    torch.manual_seed(3)
    want = torch.randperm(10).tolist()
    torch.manual_seed(3)
    first = next(iter(C.ConnectomeDataLoader(ds, batch_size=10, shuffle=True)))
    assert torch.equal(first.node_features[:10], ds[want[0]].node_features)


def test_shard_slice_partitions():
    for n in (0, 1, 7, 8, 4096):
        for w in (1, 2, 3, 8):
            parts = [shard_slice(list(range(n)), r, w) for r in range(w)]
            assert sum(parts, []) == list(range(n))
            assert max(map(len, parts)) - min(map(len, parts)) <= 1


def test_sharded_loader_covers_global_batch():
    ds = [_simple_graph(seed=i) for i in range(9)]
    full = [b for b in C.ConnectomeDataLoader(ds, batch_size=4, shuffle=False)]
    parts = [[b for b in C.ConnectomeDataLoader(ds, batch_size=4, shuffle=False, rank=r, world_size=2)]
             for r in range(2)]
    assert len(full) == 3 and len(parts[0]) == len(parts[1]) == 2   # 1-graph tail < world: dropped
    for i in range(2):
        cat = torch.cat([parts[0][i].node_features, parts[1][i].node_features])
        assert torch.equal(cat, full[i].node_features)


# ------------------------------------------------------------------------------ synthetic
def test_generator_contract():
    assert len(C.REGION_NAMES) == NUM_REGIONS == 83        # the reference's list has 83 entries
    g = C.generate_connectome(num_regions=84, seed=0)
    assert g.num_nodes == 84 and g.num_features == 5 and g.num_edges == 84 * 8
    assert (g.edge_weight > 0).all() and int(g.label) in (0, 1)
    assert int(g.edge_index.max()) < 84 and int(g.edge_index.min()) >= 0
    g2 = C.generate_connectome(num_regions=84, seed=0)
    assert torch.equal(g.edge_index, g2.edge_index) and torch.allclose(g.node_features, g2.node_features)
    assert not torch.equal(g.edge_index, C.generate_connectome(num_regions=84, seed=1).edge_index)
    s, d = g.edge_index
    assert torch.equal(s[0::2], d[1::2]) and torch.equal(d[0::2], s[1::2])       # both directions
    key = s * 84 + d
    assert key.unique().numel() == key.numel() and (s != d).all()               # simple graph
    ds = C.generate_dataset(100, num_regions=20, seed=42)
    pos = sum(int(x.label) for x in ds)
    assert len(ds) == 100 and 5 < pos < 95
    st = small_world_stats(ds[:5])
    assert set(st) == {"mean_clustering", "mean_avg_path_length", "num_graphs"} and 0 < st["mean_clustering"] < 1


def test_generator_matches_reference_distribution_g8():
    """G8 (one graph from the reference generator): same edge count, weight law and feature
    scales -- the streams differ by construction, the distributions must not."""
    d = G.load("g8_generate_seed42.npz")
    n = d["node_features"].shape[0]
    ours = C.generate_dataset(40, num_regions=n, k=8, seed=1)
    assert ours[0].num_edges == d["edge_index"].shape[1]
    w_ref = d["edge_weight"]
    w = torch.cat([g.edge_weight for g in ours]).numpy()
    assert abs(w.mean() - 2 / 7) < 0.01 and abs(float(w_ref.mean()) - 2 / 7) < 0.05      # Beta(2,5)
    x = torch.cat([g.node_features for g in ours]).numpy()
    assert np.allclose(x[:, 2].std(), 1.0, atol=0.05) and np.allclose(x[:, 4].std(), 1.0, atol=0.05)
    deg = np.bincount(ours[0].edge_index[1].numpy(), minlength=n)
    assert deg.sum() == n * 8 and deg.min() >= 2


def test_packed_dataset_equals_list_dataset():
    ds = generate_packed(6, 20, 4, seed=5)
    ls = C.generate_dataset(6, 20, 4, seed=5)
    for i in range(6):
        g = ds.graph(i)
        assert torch.equal(g.edge_index, ls[i].edge_index) and torch.equal(g.node_features, ls[i].node_features)
        assert torch.equal(g.edge_weight, ls[i].edge_weight) and int(g.label) == int(ls[i].label)


def test_resident_assemble_matches_collate_cpu():
    from connectome_gnn_amd.resident import ResidentDataLoader, assemble_batch
    ds = generate_packed(12, 20, 4, seed=5)
    ids = torch.tensor([7, 0, 3, 11])
    want = C.collate_graphs([ds.graph(int(i)) for i in ids])
    got = assemble_batch(ds, ids)
    for f in ("node_features", "edge_index", "edge_weight", "batch", "labels", "ptr"):
        assert torch.equal(getattr(got, f), getattr(want, f)), f
    ld = ResidentDataLoader(ds, batch_size=5, shuffle=False)
    assert len(ld) == 3 and [b.num_graphs for b in ld] == [5, 5, 2]


# ------------------------------------------------------------------------- tiling (fused path)
def test_tile_partition_rules():
    from connectome_gnn_amd.structure import BatchStructure
    for sizes in ([360] * 10, [84] * 512, [20] * 1000, [20, 35, 84, 7, 360, 1, 2], [384, 383, 17]):
        s = BatchStructure()
        s._ptr_host = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        s.max_nodes_per_graph = max(sizes)
        s.num_nodes = int(sum(sizes))
        s.rowptr_dst = torch.zeros(1)
        tptr, rows = s.tile_ptr(384, 256)
        cuts = tptr.tolist()
        assert cuts[0] == 0 and cuts[-1] == s.num_nodes and rows <= 384
        assert all(b > a for a, b in zip(cuts, cuts[1:]))
        assert set(cuts) <= set(s._ptr_host.tolist())          # tiles are unions of whole graphs
        assert max(b - a for a, b in zip(cuts, cuts[1:])) == rows


def test_state_dict_keys_and_param_counts():
    """identical state_dict keys/shapes and parameter counts."""
    counts = {("gcn", 64): 11234, ("sage", 64): 19746, ("gcn", 128): 42946, ("sage", 128): 76354,
              ("gcn", 256): 167810, ("sage", 256): 300162}
    for (kind, h), want in counts.items():
        m = (C.GCNConnectome if kind == "gcn" else C.GraphSAGEConnectome)(5, h)
        assert sum(p.numel() for p in m.parameters()) == want
    d = G.load("g3_models_8x20_h32.npz")
    for kind, cls in (("gcn", C.GCNConnectome), ("sage", C.GraphSAGEConnectome)):
        torch.manual_seed(42)
        sd = cls(5, 32).state_dict()
        init = G.group(d, f"{kind}_init")
        assert list(sd.keys()) == list(init.keys())
        assert all(torch.equal(sd[k], init[k]) for k in init)      # same init stream (a7/a9)
        m = cls(5, 32)
        m.load_state_dict(init)                                     # reference checkpoint loads


def test_packed_dataset_from_graphs_round_trip():
    """PackedDataset.from_graphs: the reference-style list of graphs packed for the resident loader."""
    import pytest
    import torch
    import connectome_gnn_amd as C
    from connectome_gnn_amd.synthetic import PackedDataset
    gs = C.generate_dataset(5, 20, 4, seed=3)
    ds = PackedDataset.from_graphs(gs)
    assert ds.num_subjects == 5
    for i, g in enumerate(gs):
        h = ds.graph(i)
        assert torch.equal(h.node_features, g.node_features) and torch.equal(h.edge_index, g.edge_index)
        assert torch.equal(h.edge_weight, g.edge_weight) and int(h.label) == int(g.label)
    with pytest.raises(ValueError):
        PackedDataset.from_graphs(gs + C.generate_dataset(1, 21, 4, seed=1))


def test_bench_pmc_keys_resolve_in_committed_profiles():
    """bench.py's `roofline.traffic` comes from the PMC summaries under profiles/: every workload
    that has a committed summary must resolve to a kernel in it (r2: a template parameter added
    to k_gcn_bwd silently nulled the field), and a stale prefix raises instead of returning null."""
    import importlib.util
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("cgnn_bench", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    resolved = 0
    for wl, (files, bsz, prefixes) in bench.PMC_FILES.items():
        present = [f for f in files if os.path.exists(os.path.join(root, "profiles", f))]
        if not present:
            continue
        per_launch, per_step, src = bench.pmc_traffic(wl, bsz)
        assert per_launch and per_launch > 0 and per_step and per_step > per_launch, (wl, per_launch, per_step)
        assert present[0] in src
        doc = json.load(open(os.path.join(root, "profiles", present[0])))
        assert any(k.startswith(pf) for k in doc["kernels"] for pf in prefixes)
        assert bench.pmc_traffic(wl, bsz + 1) == (None, None, None)          # another batch: no claim
        resolved += 1
    assert resolved >= 4                                # headline, cfg2, cfg3, cfg5-fp16
    bench.PMC_FILES["_stale"] = (bench.PMC_FILES["cfg4-headline-gcn-4096x360-h64"][0], 4096, ("k_no_such_kernel<",))
    with pytest.raises(RuntimeError, match="stale"):
        bench.pmc_traffic("_stale", 4096)


def test_bench_extra_configs_name_known_workloads():
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("cgnn_bench2", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    labels = [c[0] for c in bench.EXTRA_CONFIGS]
    for want in ("cfg2-gcn-512x84-h64", "cfg3-sage-512x360-h128", "cfg5-gcn-64x1000-h256-fp16",
                 "shard512-gcn-512x360-h64"):
        assert want in labels
    assert all(key in bench.WORKLOADS and launch in ("eager", "graph", "graph-dp", "auto", "trainer", "plain", "demo") for _, key, launch, _ in bench.EXTRA_CONFIGS)


def test_bench_rank_bookkeeping_for_eight_gpus():
    """The 8-GPU run nobody can rehearse on hardware: bench.py's rank bookkeeping driven for
    --gpus 8 with no GPU work -- shard sizes, launch=auto -> HIP-graph replay at 512 graphs per rank, the config
    keys that name RCCL, the unequal-shard weights, and the projected figure kept out of `value`."""
    import bench
    sizes = []
    for r in range(8):
        p = bench.plan_run("strong", "auto", 4096, r, 8)
        assert p["shard_sizes"] == [512] * 8 and p["graphs_this_rank"] == 512 and p["global_batch"] == 4096
        assert p["equal_shards"] and p["local_graphs"] is None
        assert p["launch"] == "graph"                      # 512 graphs per rank are host-bound when launched eagerly
        sizes.append(p["graphs_this_rank"])
    assert sum(sizes) == 4096
    # 1, 2, 4 GPUs of the driver's scaling sweep: eager at >= 2048 graphs per rank, replay below
    assert [bench.plan_run("strong", "auto", 4096, 0, w)["launch"] for w in (1, 2, 4, 8)] == ["eager", "eager", "graph", "graph"]
    # a global batch that does not divide: contiguous runs differing by one graph, weights = local counts
    p = [bench.plan_run("strong", "auto", 4100, r, 8) for r in range(8)]
    assert [q["graphs_this_rank"] for q in p] == [513] * 4 + [512] * 4 and not p[0]["equal_shards"]
    assert [q["local_graphs"] for q in p] == [513] * 4 + [512] * 4
    w = bench.plan_run("weak", "auto", 4096, 3, 8)
    assert w["global_batch"] == 8 * 4096 and w["graphs_this_rank"] == 4096 and w["launch"] == "eager"
    with pytest.raises(SystemExit):
        bench.plan_run("strong", "auto", 4, 0, 8)
    cfg = bench.parallel_config(8, "nccl", "strong", False, True, "split", None)
    assert cfg["backend"] == "rccl" and cfg["rccl_ranks"] == 8 and cfg["bn"] == "per-rank"
    assert cfg["launch"] == "hip-graph replay (split all-reduce)" and cfg["parallelism"] == "graph-sharded dp8"
    cfg = bench.parallel_config(8, "gloo", "strong", True, False, "split", "graph capture failed on another rank; eager launches")
    assert cfg["rccl_ranks"] == 0 and cfg["launch"] == "eager" and cfg["bn"] == "sync" and "another rank" in cfg["launch_note"]
    pr = bench.projection_8gpu(1.86, 0.327)
    assert "value" not in pr and pr["what"].startswith("projection") and pr["n_gpus"] == 8
    assert 4.5 < pr["speedup_vs_1gpu"][1] < pr["speedup_vs_1gpu"][0] < 6.0


def test_bn_tail_struct_layout_matches_header():
    """struct cgnn_bn_tail (include/cgnn.h; the library static_asserts 120 bytes) and its ctypes mirror."""
    from connectome_gnn_amd import _lib
    t = _lib.CgnnBnTail()
    assert ctypes.sizeof(t) == 120 and _lib.BN_ACC_BYTES == 16448
    assert _lib.CgnnBnTail.count.offset == 8 and _lib.CgnnBnTail.gamma.offset == 24
    assert _lib.CgnnBnTail.momentum.offset == 56 and _lib.CgnnBnTail.rng_n.offset == 88 and _lib.CgnnBnTail.bwc.offset == 112
    hdr = open(os.path.join(ROOT, "include", "cgnn.h")).read()
    assert "#define CGNN_BN_ACC_BYTES 16448" in hdr


def test_trainer_host_side_decisions_for_the_packed_path():
    """CPU-checkable pieces of Trainer's packed path (train.py): the optimizer switch, the prepare-callback
    signature probe, the claim / disarm protocol of direct gradient destinations."""
    import connectome_gnn_amd as C
    from connectome_gnn_amd import ops
    from connectome_gnn_amd.structure import call_prepare
    m = torch.nn.Linear(4, 2)
    tr = C.Trainer.__new__(C.Trainer)
    tr.optimizer = torch.optim.Adam(m.parameters(), lr=1e-3)
    assert tr._make_capturable() and all(g["capturable"] for g in tr.optimizer.param_groups)
    assert not any(g.get("fused") for g in tr.optimizer.param_groups)          # CPU parameters: no fused kernel
    tr.optimizer = torch.optim.SGD(m.parameters(), lr=1e-3)
    assert not tr._make_capturable()                                           # not an optimizer we know to switch
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    m(torch.randn(3, 4)).sum().backward()
    opt.step()
    tr.optimizer = opt
    assert not tr._make_capturable()                                           # state exists on the host already
    calls = []
    call_prepare(lambda b, reuse=False: calls.append(("kw", reuse)), 1)
    call_prepare(lambda b: calls.append(("plain",)), 1)

    def raises_inside(b, reuse=False):
        raise TypeError("from inside")
    with pytest.raises(TypeError, match="from inside"):
        call_prepare(raises_inside, 1)                                          # not swallowed, not retried
    assert calls == [("kw", True), ("plain",)]
    p = torch.nn.Parameter(torch.zeros(3))
    assert ops.grad_destination(p) is None                                      # never armed
    p._cgnn_direct = True
    p.grad = torch.zeros(3)
    assert ops.grad_destination(p) is None and p._cgnn_direct is False         # host tensors are never written to; disarmed
