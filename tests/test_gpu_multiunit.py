"""Oracle parity in the regime the benchmarks run in: a persistent workgroup that processes MORE THAN
ONE unit (tile, (tile, slice) pair, row block ...) and therefore runs its cross-unit prefetch.

Every persistent kernel walks `for (u = blockIdx.x; u < units; u += gridDim.x)` on a grid of one
workgroup per CU (256).  The parity batches of the other test files are 16 tiles / 64 units at most,
i.e. one unit per workgroup.  Two ways to reach the multi-unit regime against the oracle:

(A) the test hook cgnn_set_fused_grid(g) shrinks the grid to g in {3, 8, 16} workgroups, and the
    existing oracle suites are run again unchanged (their batches then put 2-20 units on a workgroup;
    16 also satisfies the `grid % (8 G) == 0` row-pair form of the weight-stationary GEMM, 3 its
    odd-grid fall-back and the dense aggregate's plain unit order);
(B) batches large enough for the FULL grid: GCN h64 on 300 x 360-ROI (300 tiles) and 1100 x 84-ROI
    (275 tiles of four graphs), GraphSAGE h128 on 140 x 360-ROI (280 (tile, slice) units), each with
    dropout 0.3 replayed through the oracle, on the batch and on its degree-ordered twin; the
    weight-stationary GEMMs and the half GEMMs at M = 200,000 rows.

Reference lines at stake: models.py:84-114 (GCNLayer.forward), :136-152 (SAGELayer.forward),
:203-211 (encode)."""
import itertools

import pytest
import torch

from tests import parity as P
from tests import test_gpu_kernels as K
from tests import test_gpu_models as M

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL = P.TOL
GRIDS = [3, 8, 16]


def _cases(fn):
    """The cartesian product of a test function's parametrize marks, as keyword dicts."""
    axes = []
    for mark in getattr(fn, "pytestmark", []):
        if mark.name != "parametrize":
            continue
        names = [n.strip() for n in mark.args[0].split(",")]
        rows = [v if isinstance(v, (tuple, list)) and len(names) > 1 else (v,) for v in mark.args[1]]
        axes.append([dict(zip(names, r)) for r in rows])
    out = []
    for combo in itertools.product(*axes):
        kw = {}
        for d in combo:
            kw.update(d)
        out.append(kw)
    return out


def _id(kw):
    return "-".join(str(v).replace(" ", "") for v in kw.values())


@pytest.fixture
def small_grid(request):
    from connectome_gnn_amd import _lib
    lib = _lib.load()
    full = int(lib.cgnn_fused_grid())
    assert lib.cgnn_set_fused_grid(int(request.param)) == 0
    assert int(lib.cgnn_fused_grid()) == int(request.param)
    try:
        yield int(request.param)
    finally:
        torch.cuda.synchronize()
        assert lib.cgnn_set_fused_grid(0) == 0
        assert int(lib.cgnn_fused_grid()) == full


def _suite(fn, keep=None, grids=GRIDS):
    cases = [c for c in _cases(fn) if keep is None or keep(c)]
    assert cases, fn.__name__

    @pytest.mark.parametrize("small_grid", grids, indirect=True)
    @pytest.mark.parametrize("case", cases, ids=_id)
    def run(small_grid, case):
        fn(**case)

    run.__name__ = run.__qualname__ = fn.__name__ + "_small_grid"
    run.__doc__ = f"{fn.__module__}.{fn.__name__} with 3 / 8 / 16 persistent workgroups."
    return run


# ---------------------------------------------------------------- (A) the existing oracle suites, small grid
test_models_vs_golden_small_grid = _suite(M.test_models_vs_golden)
# (the model-level suites spend their time in the CPU oracle: two grids -- 3: every workgroup takes several units,
# the odd-grid fall-backs; 16: the row-pair form of the weight-stationary GEMM -- keep the GPU run inside its budget)
test_models_vs_oracle_fresh_small_grid = _suite(M.test_models_vs_oracle_fresh, grids=[3, 16])
test_dropout_replay_small_grid = _suite(M.test_dropout_on_matches_oracle_with_replayed_masks, grids=[3, 16])
test_fused_gcn_vs_oracle_small_grid = _suite(M.test_fused_gcn_vs_oracle)
test_cfg5_fp32_small_grid = _suite(M.test_cfg5_shape_gcn_1000roi_h256_vs_oracle, lambda c: c["ngraphs"] == 5, [3, 16])
test_cfg5_fp16_small_grid = _suite(M.test_cfg5_fp16_storage_gcn_vs_fp32_oracle, grids=[3, 16])
test_sage_1000roi_small_grid = _suite(M.test_sage_1000roi_h128_band_aggregate_vs_oracle, lambda c: c["dropout"] > 0, [3, 16])
test_aggregate_tiled_small_grid = _suite(K.test_aggregate_tiled_forward_backward)
test_aggregate_tiled_bn_prologue_small_grid = _suite(K.test_aggregate_tiled_with_bn_prologue_equals_two_passes)
test_linear_small_grid = _suite(K.test_linear_forward_backward, lambda c: c["m"] >= 4096)
test_half_fwd_bwd_input_small_grid = _suite(K.test_half_storage_projection_fwd_and_bwd_input, lambda c: c["m"] >= 1000)
test_half_bwd_weight_small_grid = _suite(K.test_half_storage_projection_bwd_weight, lambda c: c["m"] >= 1000)
test_half_stats_small_grid = _suite(K.test_half_storage_projection_with_statistics_epilogue, lambda c: c["m"] >= 1000)
test_dense_aggregate_f16_small_grid = _suite(K.test_dense_aggregate_f16)
test_dense_per_fragment_small_grid = _suite(K.test_dense_per_fragment_operator_matches_dense)
test_dense_bnbwd_small_grid = _suite(K.test_dense_aggregate_with_bn_backward_prologue_equals_two_passes)
test_aggregate_tiled_f16_small_grid = _suite(K.test_aggregate_tiled_f16_storage)
test_pooled_factor_sums_small_grid = _suite(K.test_pooled_bn_pass_factor_sums_give_the_backward_statistics)


@pytest.mark.parametrize("small_grid", GRIDS, indirect=True)
def test_ws_split_product_accuracy_small_grid(small_grid):
    K.test_ws_linear_split_bf16_product_is_fp32_accurate()


@pytest.mark.parametrize("small_grid", [3, 16], indirect=True)
@pytest.mark.parametrize("kind", ["gcn", "sage"])
def test_units_per_workgroup_reached(small_grid, kind):
    """The point of (A), asserted: with g workgroups the batches above really are several units per
    workgroup (tiles for the per-tile GCN kernels, (tile, slice) pairs for the tiled aggregate)."""
    import connectome_gnn_amd as C
    b = C.collate_graphs(C.generate_dataset(6 if kind == "gcn" else 4, 360, 14, seed=321)).to(DEV)
    s = b.structure()
    meta = s.fused_meta(384, small_grid, 1.0 if kind == "gcn" else 0.0)
    tiles = int(meta.tile_ptr.numel()) - 1
    units = tiles if kind == "gcn" else tiles * 2          # hidden 128 = two 64-column slices
    assert tiles == (6 if kind == "gcn" else 4)           # one 360-ROI graph per tile whatever the grid
    if small_grid == 3:
        assert units >= 2 * small_grid                     # 6 tiles / 8 (tile, slice) units on 3 workgroups


@pytest.mark.parametrize("small_grid", [1, 3], indirect=True)
@pytest.mark.parametrize("kind,hidden", [("gcn", 64), ("sage", 128)])
def test_many_units_per_workgroup_vs_oracle(small_grid, kind, hidden):
    """The deepest walk of the persistent loops an oracle comparison reaches: 24 x 360-ROI graphs (24 one-graph
    tiles; GraphSAGE h128: 48 (tile, slice) units) on 1 or 3 workgroups -- 24 / 8 tiles (48 / 16 units) per
    workgroup, more than the 16 the headline runs -- dropout 0.3 replayed through the oracle."""
    units, grid = _full_size(kind, 360, 14, hidden, 24, False, seed=31)
    assert grid == small_grid and units == 24 * (hidden // 64)


# ---------------------------------------------------------------- (B) real sizes on the full grid
def _full_size(kind, n, k, hidden, nb, twin, seed):
    import connectome_gnn_amd as C
    from connectome_gnn_amd import _lib
    from connectome_gnn_amd.resident import assemble_batch
    from connectome_gnn_amd.synthetic import generate_packed
    grid = int(_lib.load().cgnn_fused_grid())
    ds = generate_packed(nb, n, k, seed=seed)
    b = assemble_batch(ds, torch.arange(nb))
    torch.manual_seed(13)
    m = M._model(kind, 5, hidden, dropout=0.3)
    sd0 = {k_: v.clone() for k_, v in m.state_dict().items()}
    m = m.to(DEV).train()
    m.record_dropout = True
    bd = b.to(DEV)
    if twin:
        m.prepare_batch(bd, reuse=True)
        assert bd.structure().__dict__.get("_degree_twin") is not None
    lg = m(bd)
    loss_g = torch.nn.functional.cross_entropy(lg, bd.labels)
    loss_g.backward()
    assert m.impl_used == "fused"
    s = bd.structure()
    st = s.__dict__["_degree_twin"] if twin else s       # (the twin is a BatchStructure of its own)
    meta = st.fused_meta(384, grid, 1.0 if kind == "gcn" else 0.0)
    units = (int(meta.tile_ptr.numel()) - 1) * (hidden // 64)
    assert units > grid, (units, grid)                     # some workgroup takes a second unit
    masks = P.recorded_masks(m, b.num_nodes, b.num_graphs)
    lo, loss_o, g32, st32 = P.oracle_run(kind, sd0, b, 0.3, True, masks)
    _, _, g64, _ = P.oracle_run(kind, sd0, b, 0.3, True, masks, dtype=torch.float64)
    torch.testing.assert_close(lg.detach().cpu(), lo, **TOL)
    torch.testing.assert_close(loss_g.detach().cpu(), loss_o, **TOL)
    floor = P.NoiseFloor(kind, sd0, b, 0.3, masks)
    for k_, p in m.named_parameters():
        P.assert_grad(k_, p.grad, g32[k_], g64[k_], f"fullgrid-{kind}-{nb}x{n}-h{hidden}-twin{int(twin)}", floor)
    sd = m.state_dict()
    for k_ in sd:
        if "running" in k_ or "num_batches" in k_:
            torch.testing.assert_close(sd[k_].cpu(), st32[k_], **TOL, msg=lambda s_: f"{k_}: {s_}")
    return units, grid


@pytest.mark.parametrize("twin", [False, True], ids=["batch-order", "degree-twin"])
@pytest.mark.parametrize("kind,n,k,hidden,nb", [("gcn", 360, 14, 64, 300),     # 300 one-graph tiles (headline shape)
                                               ("gcn", 84, 8, 64, 1100),      # 275 tiles of four graphs (cfg2 shape)
                                               ("sage", 360, 14, 128, 140)])  # 280 (tile, slice) units (cfg3 shape)
def test_full_grid_real_size_vs_oracle(kind, n, k, hidden, nb, twin):
    units, grid = _full_size(kind, n, k, hidden, nb, twin, seed=77)
    assert grid < units <= 2 * grid + 64


@pytest.mark.parametrize("k1,k2,n,relu", [(128, 128, 128, True),      # cfg3's [x | agg] layer: K = 256, N = 128
                                          (64, 0, 64, False),         # GCN h64 projection
                                          (128, 0, 128, False),
                                          (256, 0, 128, True)])
def test_ws_gemm_200k_rows(k1, k2, n, relu):
    """Weight-stationary fwd / bwd_input / bwd_weight at M = 200,000: ~24 row blocks of 32 per wave, against a
    float64 product (ReLU decisions within rounding of zero taken from the kernel, see K.linear_check)."""
    K.linear_check(200_000, k1, k2, n, relu, True)


@pytest.mark.parametrize("k,n", [(256, 256), (128, 128), (64, 256)])
def test_half_gemm_200k_rows(k, n):
    K.test_half_storage_projection_fwd_and_bwd_input(200_000, k, n, k)
    K.test_half_storage_projection_bwd_weight(200_000, n, k, k)
    if n in (128, 256):
        K.test_half_storage_projection_with_statistics_epilogue(200_000, k, n, k)
