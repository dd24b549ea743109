"""GPU parity of the individual HIP kernels (through the C ABI) against the oracle / plain
fp32 torch CPU references.  Integer structure is bit-exact; floating point within 1e-5
(rtol) + 1e-6 (atol) -- BASELINE.json north_star: "within 1e-5 fp32"."""
import ctypes
import math

import numpy as np
import pytest
import torch

from oracle import reference_path as O
from tests import golden_util as G

pytestmark = pytest.mark.gpu
TOL = dict(rtol=1e-5, atol=1e-6)
DEV = "cuda"


def _rand_graph_batch(sizes, deg, seed, dup=True):
    """Random block-diagonal COO with duplicates, self-loops, isolated nodes, unsorted edges."""
    g = torch.Generator().manual_seed(seed)
    srcs, dsts, off = [], [], 0
    for n in sizes:
        e = n * deg
        s = torch.randint(0, n, (e,), generator=g)
        d = torch.randint(0, n, (e,), generator=g)
        if n > 3:
            keep = (s != n - 1) & (d != n - 1) & (d != n - 2)   # n-1 isolated, n-2 no in-edges
            s, d = s[keep], d[keep]
        if dup and s.numel() > 2:
            s = torch.cat([s, s[:2]]); d = torch.cat([d, d[:2]])
        srcs.append(s + off); dsts.append(d + off); off += n
    ei = torch.stack([torch.cat(srcs), torch.cat(dsts)])
    w = torch.rand(ei.shape[1], generator=g) + 0.05
    ptr = torch.tensor([0] + list(np.cumsum(sizes)), dtype=torch.long)
    bid = torch.repeat_interleave(torch.arange(len(sizes)), torch.tensor(sizes))
    return ei, w, ptr, bid, off


def _batch(ei, w, ptr, bid, nn_, f, seed=0):
    from connectome_gnn_amd import ConnectomeBatch
    x = torch.randn(nn_, f, generator=torch.Generator().manual_seed(seed))
    return ConnectomeBatch(x, ei, w, bid, torch.zeros(len(ptr) - 1, dtype=torch.long), ptr)


def _csr_ref(ei, nn_, by):
    key = ei[by].numpy()
    order = np.argsort(key, kind="stable")
    rowptr = np.zeros(nn_ + 1, dtype=np.int64)
    np.add.at(rowptr, key + 1, 1)
    return np.cumsum(rowptr), order, ei[1 - by].numpy()[order]


def test_csr_build_long_rows_bit_exact():
    """Rows of ~60 and of > 1024 entries (the wave-per-row rank ordering and its serial tail for rows that
    do not fit its LDS keys) against numpy's stable argsort, both orderings, generic builder."""
    from connectome_gnn_amd.structure import BatchStructure
    ei, w, ptr, bid, nn_ = _rand_graph_batch([1600, 300], 60, 4)
    hub = torch.stack([torch.arange(1, 1600), torch.zeros(1599, dtype=torch.long)])     # 1599 edges into node 0
    perm = torch.randperm(ei.shape[1] + 2 * 1599, generator=torch.Generator().manual_seed(9))
    ei = torch.cat([ei, hub, hub.flip(0)], 1)[:, perm]                                   # ... and out of it
    w = torch.rand(ei.shape[1], generator=torch.Generator().manual_seed(10)) + 0.05
    b = _batch(ei, w, ptr, bid, nn_, 4).to(DEV)
    s = BatchStructure.build(b, force_generic=True)
    assert ei.shape[1] > 24 * nn_ and s.max_in_degree > 1024 and s.max_out_degree > 1024
    for by, (rp, eid, col) in ((1, (s.rowptr_dst, s.eid_dst, s.col_dst)), (0, (s.rowptr_src, s.eid_src, s.col_src))):
        rrp, reid, rcol = _csr_ref(ei, nn_, by)
        assert np.array_equal(rp.cpu().numpy(), rrp)
        assert np.array_equal(eid.cpu().numpy(), reid)
        assert np.array_equal(col.cpu().numpy(), rcol)


@pytest.mark.parametrize("sizes,deg", [([20] * 8, 4), ([5, 1, 33, 84, 2], 6), ([360, 360], 14),
                                       ([1], 0), ([700], 3), ([300, 40], 60)])
def test_csr_build_bit_exact(sizes, deg):
    ei, w, ptr, bid, nn_ = _rand_graph_batch(sizes, max(deg, 1), 1)
    if deg == 0:
        ei, w = ei[:, :0], w[:0]
    b = _batch(ei, w, ptr, bid, nn_, 4).to(DEV)
    s = b.structure()
    for by, (rp, eid, col) in ((1, (s.rowptr_dst, s.eid_dst, s.col_dst)),
                               (0, (s.rowptr_src, s.eid_src, s.col_src))):
        rrp, reid, rcol = _csr_ref(ei, nn_, by)
        assert np.array_equal(rp.cpu().numpy(), rrp)
        assert np.array_equal(eid.cpu().numpy(), reid)
        assert np.array_equal(col.cpu().numpy(), rcol)
    assert s.block_diagonal
    if ei.shape[1]:
        assert s.max_in_degree == int(np.bincount(ei[1].numpy(), minlength=nn_).max())
        assert s.max_out_degree == int(np.bincount(ei[0].numpy(), minlength=nn_).max())
    assert torch.equal(s.gptr.cpu().long(), ptr)


def test_csr_flags_cross_graph_and_range():
    from connectome_gnn_amd import ConnectomeBatch
    ei = torch.tensor([[0, 1, 2], [1, 2, 3]])
    b = ConnectomeBatch(torch.randn(4, 2), ei, torch.ones(3), torch.tensor([0, 0, 1, 1]), None,
                        torch.tensor([0, 2, 4])).to(DEV)
    assert not b.structure().block_diagonal          # edge 1->2 crosses graphs
    bad = ConnectomeBatch(torch.randn(4, 2), torch.tensor([[0, 9], [1, 2]]), torch.ones(2),
                          torch.tensor([0, 0, 1, 1]), None, torch.tensor([0, 2, 4])).to(DEV)
    with pytest.raises(IndexError):
        bad.structure()


def test_norms_match_oracle():
    ei, w, ptr, bid, nn_ = _rand_graph_batch([20, 35, 84], 5, 2)
    b = _batch(ei, w, ptr, bid, nn_, 4).to(DEV)
    s = b.structure()
    # GCN (models.py:94-108)
    ar = torch.arange(nn_)
    sa, da, wa = torch.cat([ei[0], ar]), torch.cat([ei[1], ar]), torch.cat([w, torch.ones(nn_)])
    deg = torch.zeros(nn_).scatter_add_(0, sa, wa)
    dis = (deg + 1e-8).pow(-0.5)
    coef = dis[ei[0]] * w * dis[ei[1]]
    n = s.gcn_norm()
    torch.testing.assert_close(n.dis.cpu(), dis, rtol=2e-7, atol=0)
    torch.testing.assert_close(n.selfc.cpu(), dis * dis, rtol=3e-7, atol=0)
    torch.testing.assert_close(n.coef_dst.cpu(), coef[s.eid_dst.cpu().long()], rtol=5e-7, atol=0)
    torch.testing.assert_close(n.coef_src.cpu(), coef[s.eid_src.cpu().long()], rtol=5e-7, atol=0)
    # SAGE (models.py:146-149)
    wsum = torch.zeros(nn_).scatter_add_(0, ei[1], w)
    sn = s.sage_norm()
    torch.testing.assert_close(sn.den.cpu(), wsum + 1e-8, rtol=1e-6, atol=0)
    assert torch.equal(sn.w_dst.cpu(), w[s.eid_dst.cpu().long()])
    torch.testing.assert_close(sn.coef_src_bwd.cpu(),
                               (w / (wsum + 1e-8)[ei[1]])[s.eid_src.cpu().long()], rtol=2e-6, atol=0)


@pytest.mark.parametrize("f", [1, 5, 7, 32, 64, 128, 256, 100])
def test_aggregate_forward_backward(f):
    from connectome_gnn_amd import ops
    ei, w, ptr, bid, nn_ = _rand_graph_batch([20, 35, 84, 3], 6, 3)
    b = _batch(ei, w, ptr, bid, nn_, f).to(DEV)
    s = b.structure()
    n = s.gcn_norm()
    x = b.node_features.clone().requires_grad_(True)
    bias = torch.randn(f, device=DEV, requires_grad=True)
    y = ops.aggregate(x, bias, (s.rowptr_dst, s.col_dst, n.coef_dst, n.selfc, None),
                      (s.rowptr_src, s.col_src, n.coef_src))
    cot = torch.randn(nn_, f, generator=torch.Generator().manual_seed(9))
    (y * cot.to(DEV)).sum().backward()
    # oracle: identity projection isolates the aggregation of models.py:112-114
    xc = b.node_features.cpu().clone().requires_grad_(True)
    bc = bias.detach().cpu().clone().requires_grad_(True)
    yr = O.gcn_layer(xc, ei, w, torch.eye(f), bc)
    (yr * cot).sum().backward()
    torch.testing.assert_close(y.cpu(), yr, **TOL)
    torch.testing.assert_close(x.grad.cpu(), xc.grad, **TOL)
    torch.testing.assert_close(bias.grad.cpu(), bc.grad, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("f,sizes,deg", [(64, [20, 35, 84, 3], 6), (128, [360, 7, 360], 14),
                                         (256, [100] * 9, 40), (128, [384, 1, 17], 5)])
@pytest.mark.parametrize("kind", ["gcn", "sage"])
def test_aggregate_tiled_forward_backward(f, sizes, deg, kind):
    """cgnn_aggregate_tiled_f32 (LDS-staged tiles) against the oracle's layer with an identity
    projection: GCN = dis*(A_w+I)(dis*x)+b (models.py:94-114), SAGE = A_w x/(wsum+1e-8) (:146-149)."""
    from connectome_gnn_amd import _lib, ops
    ei, w, ptr, bid, nn_ = _rand_graph_batch(sizes, deg, 11)
    b = _batch(ei, w, ptr, bid, nn_, f).to(DEV)
    s = b.structure()
    assert s.tiled_ok(f)
    grid = _lib.load().cgnn_fused_grid()
    x = b.node_features.clone().requires_grad_(True)
    cot = torch.randn(nn_, f, generator=torch.Generator().manual_seed(9))
    xc = b.node_features.cpu().clone().requires_grad_(True)
    if kind == "gcn":
        n = s.gcn_norm()
        bias = torch.randn(f, device=DEV, requires_grad=True)
        y = ops.aggregate_tiled(x, bias, s, s.fused_meta(384, grid, 1.0), pre=n.dis, post=n.dis)
        bc = bias.detach().cpu().clone().requires_grad_(True)
        yr = O.gcn_layer(xc, ei, w, torch.eye(f), bc)
    else:
        n = s.sage_norm()
        y = ops.aggregate_tiled(x, None, s, s.fused_meta(384, grid, 0.0), post=n.den, post_div=True)
        wsum = torch.zeros(nn_).scatter_add_(0, ei[1], w)
        msg = xc[ei[0]] * w[:, None]
        yr = torch.zeros(nn_, f).index_add_(0, ei[1], msg) / (wsum + 1e-8)[:, None]
    (y * cot.to(DEV)).sum().backward()
    (yr * cot).sum().backward()
    torch.testing.assert_close(y.cpu(), yr, **TOL)
    torch.testing.assert_close(x.grad.cpu(), xc.grad, **TOL)
    if kind == "gcn":
        torch.testing.assert_close(bias.grad.cpu(), bc.grad, rtol=1e-5, atol=1e-5)
    # Yadd: Y = A x + Yadd, with x and Yadd column slices of one wider buffer (SAGE backward)
    meta = s.fused_meta(384, grid, 0.0)
    wide = torch.randn(nn_, 2 * f, device=DEV)
    got = ops.aggregate_tiled_raw(s, meta, 0, wide[:, f:], None, None, None, yadd=wide[:, :f])
    plain = ops.aggregate_tiled_raw(s, meta, 0, wide[:, f:].contiguous(), None, None, None)
    torch.testing.assert_close(got, wide[:, :f] + plain, **TOL)
    # run-to-run deterministic
    assert torch.equal(plain, ops.aggregate_tiled_raw(s, meta, 0, wide[:, f:].contiguous(), None, None, None))


@pytest.mark.parametrize("sizes,deg,f", [([1000, 37, 500], 60, 128), ([84] * 5, 8, 64), ([1024], 100, 64)])
def test_aggregate_tiled_f16_storage(sizes, deg, f):
    """Config-5 scatter kernel: fp16 storage, fp32 accumulate, graphs of up to 1024 nodes and
    ~100 neighbours per node.  No fp16 reference exists -> compared with the fp32
    aggregation of the same half-rounded inputs at fp16 resolution of the result's scale."""
    from connectome_gnn_amd import _lib, ops
    ei, w, ptr, bid, nn_ = _rand_graph_batch(sizes, deg, 21)
    b = _batch(ei, w, ptr, bid, nn_, f).to(DEV)
    s = b.structure()
    grid = _lib.load().cgnn_fused_grid()
    xh = b.node_features.half()
    xr = xh.float().cpu()
    n = s.gcn_norm()
    dis = n.dis.cpu()
    for kind in ("gcn", "sage_T"):
        if kind == "gcn":
            meta = s.fused_meta(1024, grid, 1.0)
            got = ops.aggregate_tiled_f16_raw(s, meta, 0, xh, n.dis, n.dis, None)
            xs = (xr * dis[:, None]).half().float()              # the staged operand is rounded to half
            msg = xs[ei[0]] * w[:, None]
            want = (torch.zeros(nn_, f).index_add_(0, ei[1], msg) + xs) * dis[:, None]
        else:
            meta = s.fused_meta(1024, grid, 0.0)
            got = ops.aggregate_tiled_f16_raw(s, meta, ops.AGG_TRANSPOSED, xh, None, None, None)
            msg = xr[ei[1]] * w[:, None]
            want = torch.zeros(nn_, f).index_add_(0, ei[0], msg)
        scale = float(want.abs().max())
        torch.testing.assert_close(got.float().cpu(), want, rtol=2e-3, atol=2e-3 * scale)
        assert torch.equal(got, ops.aggregate_tiled_f16_raw(
            s, meta, 0 if kind == "gcn" else ops.AGG_TRANSPOSED, xh,
            n.dis if kind == "gcn" else None, n.dis if kind == "gcn" else None, None))


@pytest.mark.parametrize("sizes,deg,f", [([1000, 37, 500], 60, 128), ([84] * 5, 8, 64), ([1024], 100, 64),
                                         ([1008, 3], 30, 256)])
def test_dense_aggregate_f16(sizes, deg, f):
    """Dense per-graph operator on the fp16 matrix cores (config 5's scatter): forward (dst CSR) and
    transpose (src CSR) against the fp32 aggregation with the coefficients and inputs rounded to
    half (what the kernel is given), at fp16 resolution of the result."""
    from connectome_gnn_amd import ops
    ei, w, ptr, bid, nn_ = _rand_graph_batch(sizes, deg, 31)
    b = _batch(ei, w, ptr, bid, nn_, f).to(DEV)
    s = b.structure()
    n = s.gcn_norm()
    xh = b.node_features.half()
    xr = xh.float().cpu()
    dis = n.dis.cpu()
    c = dis[ei[0]] * w * dis[ei[1]]
    for transposed in (False, True):
        m = ops.dense_adj_f16(s, n.coef_src if transposed else n.coef_dst, n.selfc, transposed)
        got = ops.dense_aggregate_f16_raw(s, m, xh)
        src, dst = (ei[1], ei[0]) if transposed else (ei[0], ei[1])
        want = torch.zeros(nn_, f).index_add_(0, dst, xr[src] * c[:, None]) + xr * (dis * dis)[:, None]
        scale = float(want.abs().max())
        torch.testing.assert_close(got.float().cpu(), want, rtol=3e-3, atol=3e-3 * scale)
        assert torch.equal(got, ops.dense_aggregate_f16_raw(s, m, xh))
        assert torch.equal(m, ops.dense_adj_f16(s, n.coef_src if transposed else n.coef_dst, n.selfc, transposed))


def test_aggregate_deterministic():
    from connectome_gnn_amd import ops
    ei, w, ptr, bid, nn_ = _rand_graph_batch([84] * 16, 8, 4)
    b = _batch(ei, w, ptr, bid, nn_, 64).to(DEV)
    s = b.structure(); n = s.gcn_norm()
    a = (s.rowptr_dst, s.col_dst, n.coef_dst, n.selfc, None)
    t = (s.rowptr_src, s.col_src, n.coef_src)
    y1 = ops.aggregate(b.node_features, None, a, t)
    y2 = ops.aggregate(b.node_features, None, a, t)
    assert torch.equal(y1, y2)
    s2 = type(s).build(b)                      # rebuild: atomics inside must not leak into order
    assert torch.equal(s2.eid_dst, s.eid_dst) and torch.equal(s2.eid_src, s.eid_src)


@pytest.mark.parametrize("m,k1,k2,n,relu,bias", [
    (300, 5, 0, 64, False, False), (1000, 64, 0, 64, False, True), (257, 10, 10, 32, True, True),
    (129, 64, 64, 128, True, True), (64, 128, 128, 128, True, True), (77, 33, 0, 7, False, True),
    (4100, 256, 0, 256, False, False), (1, 4, 0, 7, False, True),
    # tall shapes -> the weight-stationary kernels of gemm_ws.hip (M >= 4096, N in {64,128})
    (5003, 128, 128, 128, True, True), (4097, 64, 64, 64, True, True), (6000, 128, 0, 128, False, True),
    (4500, 64, 0, 64, False, False), (8191, 64, 64, 128, True, False), (4096, 32, 0, 64, False, True),
    (4700, 256, 0, 128, False, True), (5001, 128, 0, 64, True, True)])
def test_linear_forward_backward(m, k1, k2, n, relu, bias):
    linear_check(m, k1, k2, n, relu, bias)


def linear_check(m, k1, k2, n, relu, bias):
    """ops.linear forward + all gradients against a float64 product.  A ReLU pre-activation within fp32
    accumulation noise of zero (|pre| < 1e-5: one in ~10^6 at these sizes) may be decided either way by any fp32
    evaluation, and one decision moves a whole row of dX by dy_j W[j, :]: the reference takes the kernel's own
    decisions, which must agree with float64's wherever |pre| >= 1e-5."""
    from connectome_gnn_amd import ops
    g = torch.Generator().manual_seed(m + n)
    x1 = torch.randn(m, k1, generator=g)
    x2 = torch.randn(m, k2, generator=g) if k2 else None
    w = torch.randn(n, k1 + k2, generator=g) / (k1 + k2) ** 0.5
    bv = torch.randn(n, generator=g) if bias else None
    cot = torch.randn(m, n, generator=g)
    ts = [t.clone().to(DEV).requires_grad_(True) if t is not None else None for t in (x1, x2, w, bv)]
    y = ops.linear(*ts, relu)
    (y * cot.to(DEV)).sum().backward()
    xx = (x1 if x2 is None else torch.cat([x1, x2], 1)).double()
    pre = xx @ w.double().t() + (bv.double() if bias else 0.0)
    if relu:
        keep = (y.detach().cpu() > 0)
        disagree = keep != (pre > 0)
        assert int(disagree.sum()) <= 16 and float(pre[disagree].abs().max() if disagree.any() else 0.0) < 1e-5
        want_y, dpre = pre * keep, cot.double() * keep
    else:
        want_y, dpre = pre, cot.double()

    def close(got, want):
        # reductions over M rows: 1e-5 of the tensor's scale (an element that cancels to ~0 has no
        # meaningful relative error)
        torch.testing.assert_close(got.detach().cpu().double(), want, rtol=1e-5,
                                   atol=1e-5 * float(want.abs().max()) + 1e-6)

    close(y, want_y)
    dx = dpre @ w.double()
    close(ts[0].grad, dx[:, :k1])
    if x2 is not None:
        close(ts[1].grad, dx[:, k1:])
    close(ts[2].grad, dpre.t() @ xx)
    if bias:
        close(ts[3].grad, dpre.sum(0))


def test_ws_linear_split_bf16_product_is_fp32_accurate():
    """gemm_ws.hip computes fp32 products as six bf16 partial products (split_bf16.h).  Against an
    fp64 reference the error must sit at the fp32 ACCUMULATION level (96 fp32 additions per output
    for K = 256: measured 17 x 2^-24 of sum |x||w| at worst over 590k outputs, an fmaf chain's
    size) for operands spanning six orders of magnitude, in all three kernels; a dropped second-
    order term of the split (2^-16 of a product) would show as ~250 x 2^-24."""
    from connectome_gnn_amd import ops
    g = torch.Generator().manual_seed(11)
    m, k, n = 4608, 256, 128
    mag = lambda *shape: torch.randn(*shape, generator=g) * torch.pow(10.0, torch.rand(*shape, generator=g) * 6 - 3)
    x, w, dy = mag(m, k), mag(n, k), mag(m, n)
    xd, wd, dyd = x.to(DEV), w.to(DEV), dy.to(DEV)
    y = ops.linear_fwd_raw(xd[:, :128], xd[:, 128:], wd, None, False).cpu().double()
    bound = x.double().abs() @ w.double().abs().t()
    assert float(((y - x.double() @ w.double().t()).abs() / bound).max()) < 40 * 2.0 ** -24
    dx = ops.linear_bwd_input_raw(dyd, wd, 0, k).cpu().double()
    bound = dy.double().abs() @ w.double().abs()
    assert float(((dx - dy.double() @ w.double()).abs() / bound).max()) < 40 * 2.0 ** -24
    dw = torch.empty_like(wd)
    ops.linear_bwd_weight_raw(dyd, xd, dw, 0)
    bound = dy.double().abs().t() @ x.double().abs()
    # reduction over 4608 rows in fp32: partial sums of ~18 rows per wave, then 256 partials
    assert float(((dw.cpu().double() - dy.double().t() @ x.double()).abs() / bound).max()) < 100 * 2.0 ** -24


def test_linear_a_identity_asymmetric_b():
    """MFMA layout check (guide section 3): A = I with an ASYMMETRIC B catches a transposed
    C-write that a symmetric B would hide."""
    from connectome_gnn_amd import ops
    n = 64
    w = (torch.arange(n * n, dtype=torch.float32).reshape(n, n) % 17) - 3.0
    y = ops.linear(torch.eye(n, device=DEV), None, w.to(DEV), None)
    assert torch.equal(y.cpu(), w.t())


@pytest.mark.parametrize("f", [5, 32, 64, 128, 300])
def test_pool_mean(f):
    from connectome_gnn_amd import ops
    sizes = [20, 1, 35, 84, 360]
    ptr = torch.tensor([0] + list(np.cumsum(sizes)), dtype=torch.long)
    bid = torch.repeat_interleave(torch.arange(len(sizes)), torch.tensor(sizes))
    x = torch.randn(int(ptr[-1]), f)
    xd = x.to(DEV).requires_grad_(True)
    p = ops.pool_mean(xd, ptr.to(DEV).int(), len(sizes))
    cot = torch.randn(len(sizes), f)
    (p * cot.to(DEV)).sum().backward()
    xc = x.clone().requires_grad_(True)
    pr = O.graph_mean_pool(xc, bid, len(sizes))
    (pr * cot).sum().backward()
    torch.testing.assert_close(p.cpu(), pr, **TOL)
    torch.testing.assert_close(xd.grad.cpu(), xc.grad, **TOL)


def test_cpu_tensors_raise():
    import connectome_gnn_amd as C
    g = C.generate_dataset(2, 20, 4)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        C.GCNConnectome(5, 16)(C.collate_graphs(g))


@pytest.mark.parametrize("m,n,relu,training", [(1000, 64, True, True), (257, 128, False, True),
                                               (4100, 32, True, True), (33, 256, False, True),
                                               (500, 4, True, True), (300, 64, True, False),
                                               (1, 64, False, False)])
def test_bn_act_drop_matches_torch(m, n, relu, training):
    """Layered path's fused BatchNorm(+ReLU)+dropout kernels vs torch CPU (dropout 0 = exact
    semantics: batch stats / running stats, running-stat update, all gradients)."""
    from connectome_gnn_amd import ops
    g = torch.Generator().manual_seed(m + n)
    y = torch.randn(m, n, generator=g) * 1.7 + 0.3
    cot = torch.randn(m, n, generator=g)

    def make():
        bn = torch.nn.BatchNorm1d(n)
        with torch.no_grad():
            bn.weight.copy_(torch.linspace(0.5, 1.5, n)); bn.bias.copy_(torch.linspace(-0.2, 0.3, n))
            bn.running_mean.copy_(torch.linspace(-0.1, 0.1, n)); bn.running_var.copy_(torch.linspace(0.8, 1.3, n))
        return bn.train(training)

    ref = make()
    yc = y.clone().requires_grad_(True)
    out_r = ref(yc)
    out_r = torch.relu(out_r) if relu else out_r
    (out_r * cot).sum().backward()
    mod = make().to(DEV)
    assert ops.bn_act_drop_supported(mod, n)
    yd = y.to(DEV).requires_grad_(True)
    out = ops.bn_act_drop(yd, mod, relu, 0.0, training)
    (out * cot.to(DEV)).sum().backward()
    torch.testing.assert_close(out.cpu(), out_r, rtol=1e-5, atol=2e-6)
    if m > 1 or not training:
        torch.testing.assert_close(yd.grad.cpu(), yc.grad, rtol=1e-4, atol=2e-6 + 1e-5 * float(yc.grad.abs().max()))
        for a, b in ((mod.weight.grad, ref.weight.grad), (mod.bias.grad, ref.bias.grad)):
            torch.testing.assert_close(a.cpu(), b, rtol=1e-4, atol=1e-5 * float(b.abs().max()) + 1e-6)
    torch.testing.assert_close(mod.running_mean.cpu(), ref.running_mean, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(mod.running_var.cpu(), ref.running_var, rtol=1e-5, atol=1e-6)
    assert int(mod.num_batches_tracked) == int(ref.num_batches_tracked)


def test_bn_act_drop_dropout_contract():
    from connectome_gnn_amd import ops
    y = torch.randn(20000, 64, device=DEV, requires_grad=True)
    mod = torch.nn.BatchNorm1d(64).to(DEV).train()
    out = ops.bn_act_drop(y, mod, False, 0.3, True)
    keep = (out != 0).float().mean().item()
    assert abs(keep - 0.7) < 0.01
    # E[dropout(x)] = x  (scale 1/(1-p))
    ref = torch.nn.functional.batch_norm(y.detach(), None, None, mod.weight, mod.bias, True)
    assert abs((out.detach() - ref).mean().item()) < 0.01
    out.sum().backward()
    # backward uses the forward's mask: a dropped element contributes nothing through dZ
    out2 = ops.bn_act_drop(y, mod, False, 0.3, True)
    assert not torch.equal(out2 != 0, out != 0)                  # fresh mask per call
    assert torch.isfinite(y.grad).all()


@pytest.mark.parametrize("b,c", [(4096, 2), (7, 5), (1, 2), (513, 16)])
def test_cross_entropy_matches_torch(b, c):
    from connectome_gnn_amd import ops
    g = torch.Generator().manual_seed(b + c)
    lg = (torch.randn(b, c, generator=g) * 3).requires_grad_(True)
    lab = torch.randint(0, c, (b,), generator=g)
    want = torch.nn.functional.cross_entropy(lg, lab)
    (want * 1.7).backward()
    lgd = lg.detach().to(DEV).requires_grad_(True)
    got = ops.CrossEntropyLoss()(lgd, lab.to(DEV))
    (got * 1.7).backward()
    torch.testing.assert_close(got.cpu(), want.detach(), rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(lgd.grad.cpu(), lg.grad, rtol=1e-5, atol=1e-8)


def test_cross_entropy_ignore_index_and_bad_labels():
    """torch defaults include ignore_index=-100: zero gradient, divided by the valid count.  Any
    other out-of-range label (torch raises) gives a NaN loss here."""
    from connectome_gnn_amd import ops
    g = torch.Generator().manual_seed(0)
    lg = torch.randn(37, 3, generator=g).requires_grad_(True)
    lab = torch.randint(0, 3, (37,), generator=g)
    lab[[0, 5, 36]] = -100
    want = torch.nn.functional.cross_entropy(lg, lab)
    want.backward()
    lgd = lg.detach().to(DEV).requires_grad_(True)
    got = ops.CrossEntropyLoss()(lgd, lab.to(DEV))
    got.backward()
    torch.testing.assert_close(got.cpu(), want.detach(), rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(lgd.grad.cpu(), lg.grad, rtol=1e-5, atol=1e-8)
    assert float(lgd.grad[0].abs().sum()) == 0.0
    lab[3] = 7
    assert torch.isnan(ops.CrossEntropyLoss()(lgd, lab.to(DEV)))


@pytest.mark.parametrize("h,c,p", [(64, 2, 0.0), (128, 3, 0.0), (32, 2, 0.0), (64, 2, 0.4), (256, 2, 0.0), (256, 2, 0.3)])
def test_fused_head_matches_torch(h, c, p):
    """cgnn_head_fwd/bwd against the torch modules of the reference's classifier (models.py:196-201)."""
    import torch.nn as nn
    from connectome_gnn_amd import ops
    torch.manual_seed(h + c)
    head = nn.Sequential(nn.Linear(h, h // 2), nn.ReLU(), nn.Dropout(p), nn.Linear(h // 2, c))
    x = torch.randn(777, h)
    cot = torch.randn(777, c)
    assert ops.head_supported(head)
    hd = nn.Sequential(nn.Linear(h, h // 2), nn.ReLU(), nn.Dropout(p), nn.Linear(h // 2, c)).to(DEV)
    hd.load_state_dict(head.state_dict())
    xd = x.to(DEV).requires_grad_(True)
    out = ops.head(hd, xd, training=True)
    (out * cot.to(DEV)).sum().backward()
    if p == 0.0:
        xc = x.clone().requires_grad_(True)
        ref = head.train()(xc)
        (ref * cot).sum().backward()
        torch.testing.assert_close(out.cpu(), ref, **TOL)
        torch.testing.assert_close(xd.grad.cpu(), xc.grad, rtol=1e-5, atol=1e-6)
        for (k, a), (_, b_) in zip(hd.named_parameters(), head.named_parameters()):
            torch.testing.assert_close(a.grad.cpu(), b_.grad, rtol=1e-5, atol=1e-5 * float(b_.grad.abs().max()) + 1e-7,
                                       msg=lambda s_: f"{k}: {s_}")
    else:
        # dropout contract: eval == no dropout; train output finite, a fraction ~p of hidden units dropped
        assert torch.isfinite(out).all() and torch.isfinite(xd.grad).all()
        ev = ops.head(hd, xd.detach(), training=False)
        torch.testing.assert_close(ev.cpu(), head.eval()(x), **TOL)


def test_config5_scatter_full_size_properties():
    """BASELINE config 5's scatter at full size (64 x 1000-ROI, ~100 neighbours/node, hidden 256,
    fp16 storage): the dense matrix-core form, the per-edge LDS form and the fp32 gather kernel
    agree to fp16 resolution, and the operator is linear (A(ax + y) == a A(x) + A(y))."""
    from connectome_gnn_amd import _lib, ops
    from connectome_gnn_amd.resident import assemble_batch
    from connectome_gnn_amd.synthetic import generate_packed
    ds = generate_packed(64, 1000, 100, seed=42).to(DEV)
    b = assemble_batch(ds, torch.arange(64))
    s = b.structure()
    n = s.gcn_norm()
    grid = _lib.load().cgnn_fused_grid()
    g = torch.Generator(device=DEV).manual_seed(3)
    x = torch.randn(s.num_nodes, 256, device=DEV, generator=g)
    y = torch.randn(s.num_nodes, 256, device=DEV, generator=g)
    m = ops.dense_adj_f16(s, n.coef_dst, n.selfc)
    dense = ops.dense_aggregate_f16_raw(s, m, x.half()).float()
    tiled = ops.aggregate_tiled_f16_raw(s, s.fused_meta(1024, grid, 1.0), 0, x.half(), n.dis, n.dis, None).float()
    exact = ops.aggregate_raw(s.rowptr_dst, s.col_dst, n.coef_dst, n.selfc, None, None, x.half().float())
    scale = float(exact.abs().max())
    torch.testing.assert_close(dense, exact, rtol=3e-3, atol=3e-3 * scale)
    torch.testing.assert_close(tiled, exact, rtol=3e-3, atol=3e-3 * scale)
    lin = ops.dense_aggregate_f16_raw(s, m, (0.5 * x + y).half()).float()
    want = 0.5 * dense + ops.dense_aggregate_f16_raw(s, m, y.half()).float()
    torch.testing.assert_close(lin, want, rtol=5e-3, atol=5e-3 * scale)


@pytest.mark.parametrize("sizes,deg", [([20, 35, 84, 3], 6), ([360] * 5, 14), ([1, 2, 7], 2),
                                       ([1000, 12], 9), ([84] * 40, 8)])
def test_csr_build_grouped_is_bit_identical_to_generic(sizes, deg):
    """cgnn_csr_build_grouped (one workgroup per graph, in LDS) == cgnn_csr_build on grouped COOs
    (duplicates, self-loops, isolated nodes, empty graphs); a COO that is not grouped falls back."""
    import connectome_gnn_amd as C
    from connectome_gnn_amd.structure import BatchStructure
    ei, w, ptr, bid, nn_ = _rand_graph_batch(sizes, deg, 17)
    b = _batch(ei, w, ptr, bid, nn_, 5).to(DEV)
    counts = torch.bincount(bid[ei[0]], minlength=len(sizes)) if ei.shape[1] else torch.zeros(len(sizes), dtype=torch.long)
    b._eptr = torch.cat([torch.zeros(1, dtype=torch.long), counts.cumsum(0)])
    fast, slow = BatchStructure.build(b), BatchStructure.build(b, force_generic=True)
    for name in ("rowptr_dst", "eid_dst", "col_dst", "rowptr_src", "eid_src", "col_src"):
        assert torch.equal(getattr(fast, name), getattr(slow, name)), name
    assert (fast.max_in_degree, fast.max_out_degree, fast.block_diagonal) == \
           (slow.max_in_degree, slow.max_out_degree, slow.block_diagonal)
    # collate_graphs and the resident assembler provide the grouping themselves
    empty = C.ConnectomeGraph(torch.zeros(0, 5), torch.zeros(2, 0, dtype=torch.long), torch.zeros(0),
                              torch.tensor(1))
    gs = C.generate_dataset(6, 84, 8, seed=3)
    cb = C.collate_graphs(gs[:2] + [empty] + gs[2:] + [empty]).to(DEV)
    assert cb._eptr is not None
    f2, s2 = BatchStructure.build(cb), BatchStructure.build(cb, force_generic=True)
    assert torch.equal(f2.eid_dst, s2.eid_dst) and torch.equal(f2.col_src, s2.col_src)
    # an edge that crosses graphs: flagged by the grouped kernel, rebuilt generically
    if len(sizes) > 1 and sizes[0] > 0 and sizes[1] > 0 and ei.shape[1] > 0:
        ei2 = ei.clone()
        ei2[1, 0] = sizes[0]                       # first edge now ends in graph 1
        b2 = _batch(ei2, w, ptr, bid, nn_, 5).to(DEV)
        b2._eptr = b._eptr
        s3 = BatchStructure.build(b2)
        assert not s3.block_diagonal
        assert torch.equal(s3.eid_dst, BatchStructure.build(b2, force_generic=True).eid_dst)


@pytest.mark.parametrize("wd", [0.0, 1e-4])
def test_optim_adam_matches_torch_adam(wd):
    """connectome_gnn_amd.optim.Adam (one launch, device step counter) follows torch.optim.Adam
    step for step, its state_dict loads into torch's and back, and it runs under graph capture."""
    from connectome_gnn_amd.optim import Adam
    torch.manual_seed(0)
    shapes = [(64, 5), (64,), (64, 64), (1,), (300, 7), (2, 32)]
    ps_a = [torch.randn(*s, device=DEV).requires_grad_(True) for s in shapes]
    ps_b = [p.detach().clone().requires_grad_(True) for p in ps_a]
    oa = Adam(ps_a, lr=1e-2, weight_decay=wd)
    ob = torch.optim.Adam(ps_b, lr=1e-2, weight_decay=wd)
    ps_c, ob2 = None, None
    for it in range(7):
        gs = [torch.randn_like(p) * (1.0 + it) for p in ps_a]
        for p, q, g in zip(ps_a, ps_b, gs):
            p.grad, q.grad = g.clone(), g.clone()
        if ob2 is not None:
            for r, g in zip(ps_c, gs):
                r.grad = g.clone()
            ob2.step()                   # a stock torch Adam that resumed from OUR checkpoint
        oa.step()
        ob.step()
        if it == 3:                      # checkpoint round trip through torch's format, both ways
            import copy                  # (load_state_dict aliases tensors that already fit: copy)
            sd = oa.state_dict()
            steps = [st["step"] for st in sd["state"].values()]
            assert len({t.data_ptr() for t in steps}) == len(steps), "exported step tensors are aliased"
            ps_c = [q.detach().clone().requires_grad_(True) for q in ps_b]
            ob2 = torch.optim.Adam(ps_c, lr=1e-2, weight_decay=wd)
            ob2.load_state_dict(copy.deepcopy(sd))
            assert float(ob2.state[ps_c[0]]["step"]) == 4.0
            oa.load_state_dict(copy.deepcopy(ob.state_dict()))
    for p, q in zip(ps_a, ps_b):
        torch.testing.assert_close(p, q, rtol=2e-6, atol=1e-7)
    # the resumed torch optimizer counted one step per step (not one per parameter) and followed
    assert all(float(ob2.state[r]["step"]) == 7.0 for r in ps_c)
    for r, q in zip(ps_c, ps_b):         # (it resumed from OUR moments: rounding-level differences)
        torch.testing.assert_close(r, q, rtol=1e-4, atol=1e-6)
    assert float(oa.state[ps_a[0]]["step"]) == 7.0
    for p, q in zip(ps_a, ps_b):
        torch.testing.assert_close(oa.state[p]["exp_avg_sq"], ob.state[q]["exp_avg_sq"], rtol=2e-6, atol=1e-12)
    # graph capture: three replays = three more steps
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    fixed = [torch.randn_like(p) for p in ps_a]
    for p, q, f in zip(ps_a, ps_b, fixed):
        p.grad.copy_(f)
        q.grad = f.clone()
    with torch.cuda.stream(side):
        oa.step()
    torch.cuda.current_stream().wait_stream(side)
    ob.step()
    with torch.cuda.graph(g):
        oa.step()
    for _ in range(3):
        g.replay()
        ob.step()
    torch.cuda.synchronize()
    assert float(oa.state[ps_a[0]]["step"]) == 11.0
    for p, q in zip(ps_a, ps_b):
        torch.testing.assert_close(p, q, rtol=5e-6, atol=1e-7)


@pytest.mark.parametrize("sizes,deg", [([1000, 37, 500], 60), ([84] * 5, 8), ([1024], 100), ([64, 1, 200], 3),
                                       ([1008, 3], 130), ([40, 33], 30)])
def test_dense_per_fragment_operator_matches_dense(sizes, deg):
    """cgnn_dense_pack_* + cgnn_dense_aggregate_c16 (nearly full MFMA fragments dense, the others as
    entry lists, empty ones skipped) against the dense operator: the same MFMAs accumulated
    dense-list-first, so equal up to the order of the fp32 accumulation -- and bit for bit when no
    row block mixes the two kinds.  Forward and transposed, with a bias and without."""
    from connectome_gnn_amd import ops
    ei, w, ptr, bid, nn_ = _rand_graph_batch(sizes, deg, 17)
    b = _batch(ei, w, ptr, bid, nn_, 128).to(DEV)
    s = b.structure()
    norm = s.gcn_norm()
    x = b.node_features.half()
    bias = torch.randn(128, generator=torch.Generator().manual_seed(1)).to(DEV)
    for coef, tr in ((norm.coef_dst, False), (norm.coef_src, True)):
        m = ops.dense_adj_f16(s, coef, norm.selfc, tr)
        pk = ops.dense_pack_f16(s, coef, norm.selfc, tr)
        assert pk.nnz == int((m != 0).sum())
        for bb in (None, bias):
            want = ops.dense_aggregate_f16_raw(s, m, x, bb)
            got = ops.dense_aggregate_c16_raw(s, pk, x, bb)
            if pk.num_dense == 0 or pk.num_sparse == 0:
                assert torch.equal(got, want)
            scale = float(want.float().abs().max())
            torch.testing.assert_close(got.float(), want.float(), rtol=2e-3, atol=1e-3 * scale)
            assert torch.equal(got, ops.dense_aggregate_c16_raw(s, pk, x, bb))      # deterministic
        # BatchNorm statistics in the epilogue: column sums / sums of squares of the half result
        from connectome_gnn_amd import _lib
        grid = int(_lib.load().cgnn_fused_grid())
        for fn, op in ((ops.dense_aggregate_f16_raw, m), (ops.dense_aggregate_c16_raw, pk)):
            slab = torch.full((grid, 2 * 128), float("nan"), dtype=torch.float64, device=DEV)
            y = fn(s, op, x, bias, slab)
            assert torch.equal(y, fn(s, op, x, bias))
            yd = y.double()
            torch.testing.assert_close(slab.sum(0)[:128], yd.sum(0), rtol=1e-9, atol=1e-7)
            torch.testing.assert_close(slab.sum(0)[128:], (yd * yd).sum(0), rtol=1e-9, atol=1e-7)


@pytest.mark.parametrize("sizes,deg,min_nnz", [([1000, 437, 500], 60, 28), ([1024], 100, 50), ([500, 385], 130, 0),
                                               ([1000, 3], 20, 1 << 20), ([640, 1, 400], 40, 18)])
def test_band_operator_plus_remainder_is_the_gather_aggregate(monkeypatch, sizes, deg, min_nnz):
    """cgnn_band_pack_f32 + cgnn_band_aggregate_f32 (dense fragments as exactly split bf16 MFMA products)
    followed by cgnn_aggregate_acc_f32 over the remaining edges == the gather kernel over all edges, to fp32
    rounding (each product exact to 2^-24, the sums in another order): forward and transposed, GCN
    (self-loop term, bias) and GraphSAGE (row division); every edge is in exactly one of the two parts."""
    from connectome_gnn_amd import ops
    monkeypatch.setattr(ops, "BAND_MIN_NNZ", min_nnz)
    monkeypatch.setattr(ops, "BAND_MIN_COVER", 0.0)
    ei, w, ptr, bid, nn_ = _rand_graph_batch(sizes, deg, 23)
    f = 128
    b = _batch(ei, w, ptr, bid, nn_, f).to(DEV)
    s = b.structure()
    gn, sn = s.gcn_norm(), s.sage_norm()
    x = b.node_features
    bias = torch.randn(f, generator=torch.Generator().manual_seed(1)).to(DEV)
    cases = ((s.rowptr_dst, s.col_dst, gn.coef_dst, gn.selfc, None, bias),
             (s.rowptr_src, s.col_src, gn.coef_src, gn.selfc, None, None),
             (s.rowptr_dst, s.col_dst, sn.w_dst, None, sn.den, None),
             (s.rowptr_src, s.col_src, sn.coef_src_bwd, None, None, None))
    for rowptr, col, coef, selfc, rowdiv, bb in cases:
        op = ops.band_operator_f32(s, rowptr, col, coef)
        assert op is not None
        in_band = int(coef.numel()) - int(op.coef.numel())
        assert abs(op.covered - in_band / coef.numel()) < 0.02    # (cells counted once, duplicate edges per edge)
        if min_nnz == 0:
            assert op.coef.numel() == 0                      # everything on the matrix cores
        if min_nnz >= 1 << 20:
            assert op.num_items == 0 and in_band == 0        # ... or nothing
        want = ops.aggregate_raw(rowptr, col, coef, selfc, rowdiv, bb, x)
        got = ops.aggregate_raw(rowptr, col, coef, selfc, rowdiv, bb, x, band=(s, op))
        ref = _csr_apply_f64(rowptr, col, coef, selfc, rowdiv, bb, x)
        scale = float(ref.abs().max())
        err_w, err_g = float((want.double() - ref).abs().max()), float((got.double() - ref).abs().max())
        # (a split product drops the three lowest cross terms: <= 2^-23 relative, one-sided)
        assert err_g <= max(3.0 * err_w, 1e-6 * scale), (err_g, err_w, scale)
        assert torch.equal(got, ops.aggregate_raw(rowptr, col, coef, selfc, rowdiv, bb, x, band=(s, op)))
        # a column slice of a wider buffer as destination and source
        wide = torch.zeros(nn_, 2 * f, device=DEV)
        wide[:, f:] = x
        ops.aggregate_raw(rowptr, col, coef, selfc, rowdiv, bb, wide[:, f:], out=wide[:, :f], band=(s, op))
        assert torch.equal(wide[:, :f], got)


def _csr_apply_f64(rowptr, col, coef, selfc, rowdiv, bias, x):
    n = x.shape[0]
    rows = torch.repeat_interleave(torch.arange(n, device=x.device), (rowptr[1:] - rowptr[:-1]).long())
    y = torch.zeros(n, x.shape[1], dtype=torch.float64, device=x.device)
    y.index_add_(0, rows, coef.double()[:, None] * x.double()[col.long()])
    if selfc is not None:
        y += selfc.double()[:, None] * x.double()
    if rowdiv is not None:
        y /= rowdiv.double()[:, None]
    if bias is not None:
        y += bias.double()
    return y


@pytest.mark.parametrize("h,bsz,p", [(64, 512, 0.3), (64, 37, 0.0), (128, 300, 0.3), (256, 64, 0.3), (32, 1, 0.0),
                                     (64, 4096, 0.3)])
def test_head_loss_one_launch_equals_head_then_cross_entropy(h, bsz, p):
    """cgnn_head_loss_f32 (classifier forward + mean cross-entropy + the backward of both in one launch)
    against ops.head followed by ops.cross_entropy: logits, dP and the recorded dropout factors bit for
    bit (the same arithmetic row by row, the same hash), the loss to fp32 rounding of a different fold, the
    parameter gradients to the rounding of another chunking of the row sums; ignore_index rows and an
    out-of-range label behave as in cgnn_cross_entropy_f32; a second consumer of the logits adds up."""
    from connectome_gnn_amd import ops
    g = torch.Generator().manual_seed(h + bsz)
    clf = torch.nn.Sequential(torch.nn.Linear(h, h // 2), torch.nn.ReLU(), torch.nn.Dropout(p),
                              torch.nn.Linear(h // 2, 2)).to(DEV)
    assert ops.head_loss_supported(clf)
    pooled = torch.randn(bsz, h, generator=g).to(DEV).requires_grad_(True)
    labels = torch.randint(0, 2, (bsz,), generator=g).to(DEV)
    if bsz > 8:
        labels[3] = -100
    word = torch.tensor([0x1234], dtype=torch.int32, device=DEV)
    from connectome_gnn_amd import _lib

    def run(fused, extra=False):
        for q in clf.parameters():
            q.grad = None
        pooled.grad = None
        rec = {}
        if fused:
            logits, loss = ops.head_loss(clf, pooled, labels, True, word.data_ptr(), rec)
        else:
            logits = ops.head(clf, pooled, True, word.data_ptr(), rec)
            loss = ops.cross_entropy(logits, labels)
        tot = loss + (logits * logits).sum() * 0.01 if extra else loss
        if extra:
            tot.backward()
        else:
            ops.backward_unit(loss)
        return logits.detach(), loss.detach(), pooled.grad.clone(), [q.grad.clone() for q in clf.parameters()], rec

    orig = _lib.next_seed
    try:
        _lib.next_seed = lambda dev: 0x5EED5EED12345678          # the same dropout words for both forms
        a = run(False)
        b = run(True)
        c = run(False, extra=True)
        d = run(True, extra=True)
    finally:
        _lib.next_seed = orig
    assert torch.equal(a[0], b[0]) and torch.equal(a[4]["head_factor"], b[4]["head_factor"])
    assert torch.equal(a[2], b[2])                       # dP: row-local, the same chain
    torch.testing.assert_close(b[1], a[1], rtol=2e-6, atol=1e-7)
    for x, y in zip(a[3], b[3]):
        torch.testing.assert_close(y, x, rtol=2e-5, atol=2e-7)
    torch.testing.assert_close(d[2], c[2], rtol=2e-5, atol=1e-7)
    for x, y in zip(c[3], d[3]):
        torch.testing.assert_close(y, x, rtol=2e-5, atol=2e-6)
    if bsz > 8:
        assert float(b[2][3].abs().max()) == 0.0         # ignored row: no gradient
        labels[5] = 7                                    # torch raises; the kernels turn the loss into NaN
        assert torch.isnan(run(True)[1]) and torch.isnan(run(False)[1])


@pytest.mark.parametrize("p", [0.0, 0.3])
def test_aggregate_tiled_with_bn_prologue_equals_two_passes(p):
    """cgnn_aggregate_tiled_bn_f32 (BatchNorm + dropout of the input applied while the tiles are
    staged, X' and its keep bytes written on the side) == cgnn_bn_act_fwd_apply followed by
    cgnn_aggregate_tiled_f32, bit for bit."""
    from connectome_gnn_amd import _lib, ops
    ei, w, ptr, bid, nn_ = _rand_graph_batch([360, 7, 360, 84], 14, 5)
    f = 128
    b = _batch(ei, w, ptr, bid, nn_, f).to(DEV)
    s = b.structure()
    lib = _lib.load()
    ell = s.fused_meta(384, int(lib.cgnn_fused_grid()), 0.0)
    norm = s.sage_norm(backward_coef=False)
    g = torch.Generator().manual_seed(2)
    z = torch.randn(nn_, f, generator=g).to(DEV)
    coef = torch.randn(4 * f, generator=g).to(DEV)
    seed = 0x1234567890ABCDEF
    x_ref = torch.empty_like(z)
    m_ref = torch.zeros(nn_ * f // 4, dtype=torch.uint8, device=DEV)
    _lib.check(lib.cgnn_bn_act_fwd_apply(_lib.ptr(z), _lib.ptr(coef), 0, p, seed, None, _lib.ptr(m_ref) if p else None,
                                         _lib.ptr(x_ref), nn_, f, _lib.stream_ptr()), "apply")
    y_ref = ops.aggregate_tiled_raw(s, ell, ops.AGG_POST_DIV, x_ref, None, norm.den, None)
    x = torch.full_like(z, float("nan"))
    m = torch.zeros_like(m_ref)
    y = ops.aggregate_tiled_bn_raw(s, ell, ops.AGG_POST_DIV, z, None, norm.den, None, coef, False, p, seed, None,
                                   m if p else None, x)
    assert torch.equal(x, x_ref) and torch.equal(y, y_ref) and torch.equal(m, m_ref)


@pytest.mark.parametrize("m,k,n,kw", [(64000, 256, 256, 256), (1000, 256, 256, 256), (4097, 128, 128, 128),
                                      (33, 256, 64, 256), (2000, 64, 256, 5), (777, 64, 128, 64), (31, 32, 64, 32)])
def test_half_storage_projection_fwd_and_bwd_input(m, k, n, kw):
    """cgnn_linear_fwd_f16 / cgnn_linear_bwd_input_f16 (gemm_h16.hip: weight-stationary, half
    activations, fp32 weights converted in the kernel, fp32 accumulate) against a float64 product
    of the SAME half-rounded operands; the result is that product rounded once to half."""
    from connectome_gnn_amd import ops
    g = torch.Generator().manual_seed(m + k + n)
    x = (torch.randn(m, k, generator=g) * 0.7).half().to(DEV)
    w = (torch.randn(n, kw, generator=g) / math.sqrt(kw)).to(DEV)
    b = torch.randn(n, generator=g).to(DEV)
    wh = w.half().double()
    y = ops.linear_fwd_f16_raw(x, w, b)
    want = x[:, :kw].double() @ wh.t() + b.double()
    assert y.dtype == torch.float16 and y.shape == (m, n)
    err = (y.double() - want).abs().max().item()
    assert err <= 1.5e-3 * want.abs().max().item(), err            # one rounding to half (2^-11 relative)
    if kw == k and k in (64, 128, 256):
        dy = (torch.randn(m, n, generator=g) * 0.5).half().to(DEV)
        dx = ops.linear_bwd_input_f16_raw(dy, w)
        want = dy.double() @ wh
        assert dx.shape == (m, k)
        err = (dx.double() - want).abs().max().item()
        assert err <= 1.5e-3 * want.abs().max().item(), err
    # determinism: the same launch twice is bit-identical
    assert torch.equal(y, ops.linear_fwd_f16_raw(x, w, b))


@pytest.mark.parametrize("m,n,k,kw", [(64000, 256, 256, 256), (64000, 256, 64, 5), (1000, 128, 128, 128),
                                      (4099, 256, 128, 100), (37, 128, 64, 64), (1, 256, 256, 256),
                                      (5000, 64, 64, 64), (3001, 64, 128, 128), (900, 64, 256, 200)])
def test_half_storage_projection_bwd_weight(m, n, k, kw):
    """cgnn_linear_bwd_weight_f16 (dW = dY^T X with both operands read column-wise out of LDS by
    ds_read_b64_tr_b16, fp32 partials per run of rows folded in fixed order) against float64."""
    from connectome_gnn_amd import ops
    g = torch.Generator().manual_seed(3 * m + n + k)
    dy = (torch.randn(m, n, generator=g) * 0.5).half().to(DEV)
    x = (torch.randn(m, k, generator=g) * 0.7).half().to(DEV)
    dw = ops.linear_bwd_weight_f16_raw(dy, x, kw)
    want = dy.double().t() @ x[:, :kw].double()
    assert dw.dtype == torch.float32 and dw.shape == (n, kw)
    err = (dw.double() - want).abs().max().item()
    assert err <= 2e-6 * max(want.abs().max().item(), 1.0) * math.sqrt(m), err     # fp32 accumulation over m rows
    assert torch.equal(dw, ops.linear_bwd_weight_f16_raw(dy, x, kw))               # fixed order: bit-identical
    xf = torch.randn(9, 5, generator=g).to(DEV)
    pc = ops.pad_cast_f16(xf, 64)
    assert pc.shape == (9, 64) and torch.equal(pc[:, :5], xf.half()) and not pc[:, 5:].any()


@pytest.mark.parametrize("m,k,n,kw", [(64000, 64, 256, 5), (4097, 128, 128, 128), (1000, 256, 256, 256), (31, 64, 128, 64)])
def test_half_storage_projection_with_statistics_epilogue(m, k, n, kw):
    """cgnn_linear_fwd_stats_f16: the same output as cgnn_linear_fwd_f16 bit for bit, and the column
    sums / sums of squares of that (half-rounded) output in the per-workgroup fp64 slab."""
    from connectome_gnn_amd import _lib, ops
    g = torch.Generator().manual_seed(m + k + n)
    x = (torch.randn(m, k, generator=g) * 0.7 + 0.2).half().to(DEV)
    w = (torch.randn(n, kw, generator=g) / math.sqrt(kw)).to(DEV)
    b = torch.randn(n, generator=g).to(DEV)
    grid = int(_lib.load().cgnn_fused_grid())
    y, slab = ops.linear_fwd_stats_f16_raw(x, w, b, grid)
    assert y is not None and slab.shape == (grid, 2 * n)
    assert torch.equal(y, ops.linear_fwd_f16_raw(x, w, b))
    s = slab.sum(dim=0)
    yd = y.double()
    torch.testing.assert_close(s[:n], yd.sum(dim=0), rtol=1e-12, atol=1e-9)
    torch.testing.assert_close(s[n:], (yd * yd).sum(dim=0), rtol=1e-12, atol=1e-9)
    assert ops.linear_fwd_stats_f16_raw(x, w[:64], b[:64], grid) == (None, None)      # N = 64: not covered


@pytest.mark.parametrize("sizes,n,relu,p,half", [([1000, 37, 500], 256, True, 0.3, True), ([84] * 9, 64, False, 0.3, False),
                                                 ([360, 1, 200], 128, True, 0.0, False), ([5] * 300, 64, False, 0.5, True)])
def test_pooled_bn_pass_factor_sums_give_the_backward_statistics(sizes, n, relu, p, half):
    """cgnn_bn_act_pool_fwd(Fsum) + cgnn_bn_act_pool_bwd_finalize == the dP form of
    cgnn_bn_act_bwd_stats + cgnn_bn_act_bwd_finalize (the pass over Y they replace), and the pooled
    rows / keep bytes are those of the pass without factor sums."""
    from connectome_gnn_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(len(sizes) + n)
    m, bsz = sum(sizes), len(sizes)
    y = (torch.randn(m, n, generator=g) * 1.3 + 0.2).to(DEV)
    y = y.half() if half else y
    coef = torch.cat([torch.rand(n, generator=g) + 0.5, torch.randn(n, generator=g) * 0.3,
                      torch.randn(n, generator=g) * 0.2, torch.rand(n, generator=g) + 0.7]).to(DEV)
    gptr = torch.tensor([0] + list(np.cumsum(sizes)), dtype=torch.int32, device=DEV)
    node_graph = torch.repeat_interleave(torch.arange(bsz, dtype=torch.int32), torch.tensor(sizes)).to(DEV)
    dP = torch.randn(bsz, n, generator=g).to(DEV)
    sfx = "_f16" if half else ""
    sp = _lib.stream_ptr(torch.device(DEV))
    outs = []
    for with_f in (False, True):
        pooled = torch.empty(bsz, n, device=DEV)
        mask = torch.zeros(m * (n // 4), dtype=torch.uint8, device=DEV)
        fsum = torch.empty(2, bsz, n, device=DEV) if with_f else None
        _lib.check(getattr(lib, "cgnn_bn_act_pool_fwd" + sfx)(_lib.ptr(y), _lib.ptr(coef), int(relu), p, 1234, None,
                                                             _lib.ptr(mask), _lib.ptr(gptr), bsz, _lib.ptr(pooled), n,
                                                             _lib.ptr(fsum), sp), "pool_fwd")
        outs.append((pooled, mask, fsum))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    _, mask, fsum = outs[1]
    got = [torch.empty(n, device=DEV), torch.empty(n, device=DEV), torch.empty(2 * n, device=DEV)]
    _lib.check(lib.cgnn_bn_act_pool_bwd_finalize(_lib.ptr(dP), _lib.ptr(fsum), _lib.ptr(gptr), bsz, n, float(m), 0,
                                                 *(_lib.ptr(t) for t in got), sp), "pool_bwd_finalize")
    rows = int(lib.cgnn_bn_act_slab_rows(m))
    slab = torch.empty(rows, 2 * n, dtype=torch.float64, device=DEV)
    _lib.check(getattr(lib, "cgnn_bn_act_bwd_stats" + sfx)(None, _lib.ptr(y), _lib.ptr(mask), _lib.ptr(coef), int(relu), p,
                                                          m, n, _lib.ptr(slab), _lib.nbytes(slab), _lib.ptr(dP), _lib.ptr(node_graph),
                                                          _lib.ptr(gptr), sp), "bwd_stats")
    want = [torch.empty(n, device=DEV), torch.empty(n, device=DEV), torch.empty(2 * n, device=DEV)]
    _lib.check(lib.cgnn_bn_act_bwd_finalize(_lib.ptr(slab), rows, n, float(m), None, 0, *(_lib.ptr(t) for t in want), sp),
               "bwd_finalize")
    for a, b in zip(got, want):
        torch.testing.assert_close(a, b, rtol=2e-5, atol=2e-5 * float(b.abs().max()) + 1e-7)


def test_gather_rows_and_multi_reduce():
    """cgnn_gather_rows (several row gathers by one id list, 16- and 4-byte paths) and
    cgnn_slab_reduce_f64_multi (several slab folds in one launch) against torch indexing / sums."""
    from connectome_gnn_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(5)
    s = 37
    srcs = [torch.randn(s, 84, 5, generator=g).to(DEV), torch.randint(0, 9, (s,), generator=g).to(DEV),
            torch.randint(0, 1 << 20, (s, 7), generator=g, dtype=torch.int32).to(DEV), torch.randn(s, 84, generator=g).to(DEV)]
    ids = torch.tensor([3, 3, 36, 0, 17, 5, 36], dtype=torch.long, device=DEV)
    dsts = [torch.empty((ids.numel(),) + tuple(t.shape[1:]), dtype=t.dtype, device=DEV) for t in srcs]
    jobs = _lib.CgnnGatherJobs()
    jobs.n = len(srcs)
    for i, (a, b) in enumerate(zip(srcs, dsts)):
        jobs.src[i], jobs.dst[i], jobs.row_bytes[i] = a.data_ptr(), b.data_ptr(), a[0].numel() * a.element_size()
    sp = _lib.stream_ptr(torch.device(DEV))
    _lib.check(lib.cgnn_gather_rows(jobs, _lib.ptr(ids), ids.numel(), None, sp), "gather_rows")
    for a, b in zip(srcs, dsts):
        assert torch.equal(b, a.index_select(0, ids))
    # the id list as a window of a longer one, its position read from a device cursor
    cursor = torch.tensor([2], dtype=torch.long, device=DEV)
    tally = torch.tensor([0.5], device=DEV)
    _lib.check(lib.cgnn_gather_rows(jobs, _lib.ptr(ids), 4, _lib.ptr(cursor), sp), "gather_rows")
    for a, b in zip(srcs, dsts):
        assert torch.equal(b[:4], a.index_select(0, ids[2:6]))
    _lib.check(lib.cgnn_epoch_advance(_lib.ptr(cursor), 4, _lib.ptr(torch.tensor([0.25], device=DEV)), 8.0, _lib.ptr(tally), sp),
               "epoch_advance")
    assert int(cursor) == 6 and float(tally) == 2.5
    jobs.row_bytes[1] = 6
    assert lib.cgnn_gather_rows(jobs, _lib.ptr(ids), ids.numel(), None, sp) == _lib.CGNN_EINVAL   # not a multiple of 4
    red = _lib.DeferredReduce()
    slabs = [torch.randn(r, w, generator=g, dtype=torch.float64).to(DEV) for r, w in ((1024, 256), (3, 64), (250, 128))]
    outs = [torch.empty(t.shape[1], device=DEV) for t in slabs]
    for t, o in zip(slabs, outs):
        red.add(t, t.shape[0], t.shape[1], o)
    red.flush(sp)
    for t, o in zip(slabs, outs):
        torch.testing.assert_close(o, t.sum(dim=0).float(), rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("sizes,deg,f,p,pooled", [([1000, 37, 500], 60, 256, 0.3, False), ([84] * 5, 8, 64, 0.0, False),
                                                   ([1008, 3], 130, 128, 0.3, True), ([64, 1, 200], 3, 64, 0.5, True)])
def test_dense_aggregate_with_bn_backward_prologue_equals_two_passes(sizes, deg, f, p, pooled):
    """cgnn_dense_aggregate_c16_bnbwd (dY = BatchNorm'(dX' * relu' * drop') formed while the slices are
    staged, db left per graph) == cgnn_bn_act_bwd_apply_f16 followed by cgnn_dense_aggregate_c16: the
    same half-rounded dY goes through the same MFMAs (bit-identical dT), the bias gradient agrees to
    the order of its fp64 fold."""
    from connectome_gnn_amd import _lib, ops
    lib = _lib.load()
    ei, w, ptr, bid, nn_ = _rand_graph_batch(sizes, deg, 23)
    b = _batch(ei, w, ptr, bid, nn_, f).to(DEV)
    s = b.structure()
    norm = s.gcn_norm()
    pk = ops.dense_pack_f16(s, norm.coef_src, norm.selfc, True)
    g = torch.Generator().manual_seed(f + len(sizes))
    y = (torch.randn(nn_, f, generator=g) * 1.2 + 0.1).half().to(DEV)
    dx = None if pooled else (torch.randn(nn_, f, generator=g) * 0.3).half().to(DEV)
    dP = torch.randn(len(sizes), f, generator=g).to(DEV) if pooled else None
    coef = torch.cat([torch.rand(f, generator=g) + 0.5, torch.randn(f, generator=g) * 0.3,
                      torch.randn(f, generator=g) * 0.2, torch.rand(f, generator=g) + 0.7]).to(DEV)
    bwc = (torch.randn(2 * f, generator=g) * 0.05).to(DEV)
    mask = torch.randint(0, 16, (nn_ * f // 4,), generator=g, dtype=torch.uint8).to(DEV) if p > 0 else None
    sp = _lib.stream_ptr(torch.device(DEV))
    rows = int(lib.cgnn_bn_act_apply_blocks(nn_, f))
    cs = torch.empty(rows, f, dtype=torch.float64, device=DEV)
    dy = torch.empty_like(y)
    pool = (_lib.ptr(dP), _lib.ptr(s.node_graph), _lib.ptr(s.gptr)) if pooled else (None, None, None)
    _lib.check(lib.cgnn_bn_act_bwd_apply_f16(_lib.ptr(dx), _lib.ptr(y), _lib.ptr(mask), _lib.ptr(coef), _lib.ptr(bwc), 1, p, 0,
                                             _lib.ptr(cs), _lib.nbytes(cs), _lib.ptr(dy), nn_, f, *pool, sp), "bwd_apply")
    want = ops.dense_aggregate_c16_raw(s, pk, dy)
    got, cs2 = ops.dense_aggregate_c16_bnbwd_raw(s, pk, dx, dP, y, mask, coef, bwc, True, p)
    assert cs2.shape == (len(sizes), f)
    assert torch.equal(got, want)
    db_want, db_got = cs.sum(0), cs2.sum(0)
    torch.testing.assert_close(db_got, db_want, rtol=1e-6, atol=1e-6 * float(dy.float().abs().sum(0).max()) + 1e-9)


# ------------------------------------------------------------------ ABI 2: scratch buffers carry their size
def test_short_scratch_buffers_are_refused_before_any_launch():
    """Every scratch / partial-sum buffer the library writes is passed with its byte count (ABI 2).  The
    round-3 fault -- GraphSAGE hidden 256 on >= 4096 nodes wrote 33 MB of weight-gradient partials into a
    1 MB slab -- is now CGNN_EINVAL (-1): a 1-byte slab at that shape, and at shapes of the other slab
    writers, is refused and nothing is launched (the outputs keep their sentinel)."""
    from connectome_gnn_amd import _lib
    lib = _lib.load()
    sp = _lib.stream_ptr()
    m, hid = 5760, 256                                        # 16 x 360-ROI graphs, GraphSAGE hidden 256
    dy = torch.randn(m, hid, device=DEV)
    x1, x2 = torch.randn(m, hid, device=DEV), torch.randn(m, hid, device=DEV)
    dw = torch.full((hid, 2 * hid), 7.0, device=DEV)
    tiny = torch.zeros(1, dtype=torch.uint8, device=DEV)
    need = int(lib.cgnn_linear_bwd_weight2_workspace_bytes(m, hid, hid, hid))
    assert need > 8 << 20                                      # the per-panel weight-stationary form: 256 partials
    rc = lib.cgnn_linear_bwd_weight2_f32(_lib.ptr(dy), hid, _lib.ptr(x1), hid, hid, _lib.ptr(x2), hid, hid,
                                         _lib.ptr(dw), 2 * hid, m, hid, _lib.ptr(tiny), 1, sp)
    assert rc == _lib.CGNN_EINVAL
    rc = lib.cgnn_linear_bwd_weight_f32(_lib.ptr(dy), hid, _lib.ptr(x1), hid, _lib.ptr(dw), 2 * hid, 0, m, hid, hid,
                                        _lib.ptr(tiny), 1, sp)
    assert rc == _lib.CGNN_EINVAL
    # one byte short of the documented size is refused too, the documented size is accepted
    ws = torch.empty(need, dtype=torch.uint8, device=DEV)
    assert lib.cgnn_linear_bwd_weight2_f32(_lib.ptr(dy), hid, _lib.ptr(x1), hid, hid, _lib.ptr(x2), hid, hid,
                                           _lib.ptr(dw), 2 * hid, m, hid, _lib.ptr(ws), need - 257, sp) == _lib.CGNN_EINVAL
    torch.cuda.synchronize()
    assert bool((dw == 7.0).all())                             # nothing ran
    assert lib.cgnn_linear_bwd_weight2_f32(_lib.ptr(dy), hid, _lib.ptr(x1), hid, hid, _lib.ptr(x2), hid, hid,
                                           _lib.ptr(dw), 2 * hid, m, hid, _lib.ptr(ws), need, sp) == _lib.CGNN_OK
    want = dy.double().t() @ torch.cat([x1, x2], 1).double()
    torch.testing.assert_close(dw.double(), want, rtol=1e-5, atol=1e-5 * float(want.abs().max()))
    # the other families of slab writers
    out = torch.full((hid,), 7.0, device=DEV)
    assert lib.cgnn_colsum_f32(_lib.ptr(dy), hid, _lib.ptr(out), m, hid, _lib.ptr(tiny), 1, sp) == _lib.CGNN_EINVAL
    assert lib.cgnn_bn_act_fwd_stats(_lib.ptr(dy), m, hid, _lib.ptr(tiny), 1, sp) == _lib.CGNN_EINVAL
    assert lib.cgnn_bn_act_bwd_apply(_lib.ptr(dy), _lib.ptr(x1), None, _lib.ptr(x2), _lib.ptr(x2), 1, 0.0, 0,
                                     _lib.ptr(tiny), 1, _lib.ptr(dw), m, hid, None, None, None, sp) == _lib.CGNN_EINVAL
    dyh = dy.half()
    assert lib.cgnn_linear_bwd_weight_f16(_lib.ptr(dyh), hid, _lib.ptr(dyh), hid, _lib.ptr(dw), 2 * hid, hid, m, hid,
                                          hid, _lib.ptr(tiny), 1, sp) == _lib.CGNN_EINVAL
    assert lib.cgnn_linear_fwd_stats_f16(_lib.ptr(dyh), hid, hid, _lib.ptr(x1), hid, hid, None, _lib.ptr(dyh), hid, m,
                                         hid, _lib.ptr(tiny), 1, sp) in (_lib.CGNN_EINVAL, _lib.CGNN_EUNSUPPORTED)
    h2, c, bsz = 32, 2, 512
    z = torch.zeros(bsz * 64, device=DEV)
    assert lib.cgnn_head_bwd_f32(_lib.ptr(z), _lib.ptr(z), _lib.ptr(z), _lib.ptr(z), bsz, 64, h2, c, _lib.ptr(z),
                                 _lib.ptr(z), _lib.ptr(z), _lib.ptr(tiny), 1, sp) == _lib.CGNN_EINVAL
    b = _batch(*_rand_graph_batch([84] * 6, 8, 3), 5).to(DEV)
    s = b.structure()
    grid = int(lib.cgnn_fused_grid())
    meta = s.fused_meta(384, grid, 1.0)
    tp = ctypes.byref(s.tiles_struct(meta, s.gcn_dis(meta)))
    y = torch.full((b.num_nodes, 64), 7.0, device=DEV)
    w0, b0 = torch.randn(64, 5, device=DEV), torch.zeros(64, device=DEV)
    assert lib.cgnn_gcn_fused_fwd_first(tp, _lib.ptr(b.node_features), 5, _lib.ptr(w0), _lib.ptr(b0), _lib.ptr(y),
                                        _lib.ptr(tiny), 1, sp) == _lib.CGNN_EINVAL
    torch.cuda.synchronize()
    assert bool((y == 7.0).all()) and bool((out == 7.0).all())


@pytest.mark.parametrize("m,n,k1,k2", [(5000, 128, 64, 256), (4500, 128, 256, 64), (4100, 64, 32, 128), (300, 128, 64, 256),
                                       (5000, 256, 64, 128)])
def test_two_panel_weight_gradient_with_unequal_panels(m, n, k1, k2):
    """cgnn_linear_bwd_weight2_f32 with K1 != K2: outside the joint weight-stationary form each panel may take its
    own (one [N x Ki] partial per workgroup); the slab is sized by cgnn_linear_bwd_weight2_workspace_bytes from
    the actual panels (sizing from K1 + K2 under-counted 64 + 256 at N = 128, M >= 4096)."""
    from connectome_gnn_amd import _lib, ops
    lib = _lib.load()
    g = torch.Generator().manual_seed(m + n + k1)
    dy, x1, x2 = torch.randn(m, n, generator=g), torch.randn(m, k1, generator=g), torch.randn(m, k2, generator=g)
    need = int(lib.cgnn_linear_bwd_weight2_workspace_bytes(m, n, k1, k2))
    for k in (k1, k2):
        assert need >= int(lib.cgnn_linear_bwd_weight_workspace_bytes(m, n, k))
    dw = torch.empty(n, k1 + k2, device=DEV)
    ops.linear_bwd_weight2_raw(dy.to(DEV), x1.to(DEV), x2.to(DEV), dw)
    want = dy.double().t() @ torch.cat([x1, x2], 1).double()
    torch.testing.assert_close(dw.cpu().double(), want, rtol=1e-5, atol=1e-5 * float(want.abs().max()) + 1e-6)


def test_pool_mean_more_graphs_than_a_grid_dimension():
    """Mean-pool readout (models.py:40-47,57-59) over more than 65535 graphs (whole-dataset evaluation of small
    graphs): the graphs ride on gridDim.y in runs of 65535."""
    from connectome_gnn_amd import ops
    B, f = 70_001, 8
    sizes = torch.randint(1, 4, (B,), generator=torch.Generator().manual_seed(2))
    ptr = torch.cat([torch.zeros(1, dtype=torch.long), torch.cumsum(sizes, 0)])
    nn_ = int(ptr[-1])
    x = torch.randn(nn_, f, generator=torch.Generator().manual_seed(3))
    xd = x.to(DEV).requires_grad_(True)
    p = ops.pool_mean(xd, ptr.to(device=DEV, dtype=torch.int32), B)
    seg = torch.repeat_interleave(torch.arange(B), sizes)
    want = torch.zeros(B, f).index_add_(0, seg, x) / (sizes.float() + 1e-8)[:, None]
    torch.testing.assert_close(p.cpu(), want, **TOL)
    cot = torch.randn(B, f, generator=torch.Generator().manual_seed(4))
    (p * cot.to(DEV)).sum().backward()
    torch.testing.assert_close(xd.grad.cpu(), (cot / (sizes.float() + 1e-8)[:, None])[seg], **TOL)
