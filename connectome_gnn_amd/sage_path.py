"""GraphSAGE encoder as one autograd node (hidden % 64 == 0): host orchestration.

GraphSAGEConnectome.encode (reference models.py:256-262) through the generic ops is ~25 autograd
nodes per step; autograd then inserts a ReLU-mask multiply, a gradient add and a column sum per
layer as separate passes over [Nn, H] arrays.  Here the whole encoder is one
``torch.autograd.Function`` whose backward is written out by hand, so that every pass over a node
array is a HIP kernel that does several things at once:

  forward, layer l     A  = A_w X / (wsum + 1e-8)            cgnn_aggregate_tiled_f32 (LDS tiles)
                       Z  = relu([X | A] W^T + b)            cgnn_linear_fwd_f32 (MFMA, W in LDS)
                       X' = dropout(BatchNorm(Z))            cgnn_bn_act_* (stats, finalize, apply)
  backward, layer l    dPre = BatchNorm'(dX' * drop') * (Z > 0), db = colsum(dPre)
                                                             cgnn_bn_act_bwd_* (one apply pass)
                       dW = dPre^T [X | A]                   cgnn_linear_bwd_weight2_f32 (one pass)
                       [dX1 | dA] = dPre W                   cgnn_linear_bwd_input_f32 (one pass)
                       dX = dX1 + A_w^T (dA / (wsum+1e-8))   cgnn_aggregate_tiled_f32 (+Yadd)

No arithmetic of the path happens in torch here; torch allocates buffers and orders the launches.
"""
from __future__ import annotations

from typing import List, Optional

import torch

from . import _lib, ops
from .structure import BatchStructure

TILE_ROWS = 384


def eligible(model, batch, structure: BatchStructure) -> Optional[str]:
    """None if this path covers (model, batch); else the reason it does not."""
    hid = model.convs[0].linear.weight.shape[0]
    if hid % 64 or not bool(_lib.load().cgnn_bn_act_width_ok(hid)):
        return "hidden_dim is not 64, 128, 256, ..."
    if not structure.tiled_ok(hid) and not isinstance(structure, BatchStructure):
        return "graphs do not fit an LDS tile and the structure has no CSR form"
    if not structure.tiled_ok(hid) and hid not in (64, 128, 256):
        # off the tiled aggregate the backward adds dX1 inside the gather kernel (ops.aggregate_raw(yadd=...)),
        # which covers widths 64 / 128 / 256 only: wider layers on large graphs take the layered path
        return "graphs do not fit an LDS tile and hidden_dim is not 64, 128 or 256"
    if batch.node_features.requires_grad:
        return "node_features require grad"
    if not bn_modules_ok(model):
        return "BatchNorm is not a plain affine BatchNorm1d / SyncBatchNorm with running stats"
    return None


class _Saved:
    __slots__ = ("s", "ell", "norm", "xs", "aggs", "zs", "coefs", "masks", "p", "training", "ws", "xa0",
                 "sync_group", "count_block", "fsum", "tiled", "band", "grad_dst")


PAD_K = 32          # layer 0: [x0 | agg(x0) | 0] packed to one 32-wide panel
PAD_MIN_ROWS = 4096  # (= the row count from which the weight-stationary GEMMs apply)


def _f32(dev, *shape):
    return torch.empty(*shape, dtype=torch.float32, device=dev)


def _agg_narrow_tiled(s, ell, norm, x, out=None):
    """The same mean for a narrow x (layer 0's few input features) through the LDS-tiled aggregate on a
    zero-padded 64-column panel -- for structures that carry no CSR (the per-subject structure cache)."""
    f = x.shape[1]
    panel = torch.nn.functional.pad(x, (0, 64 * ((f + 63) // 64) - f))
    agg = ops.aggregate_tiled_raw(s, ell, ops.AGG_POST_DIV, panel, None, norm.den, None)[:, :f]
    if out is None:
        return agg.contiguous()
    out.copy_(agg)
    return out


def _agg_fwd(s: BatchStructure, ell, norm, x, band=None):
    """weighted mean of in-neighbours, models.py:146-149"""
    if ell is not None and s.tiled_ok(x.shape[1]):
        return ops.aggregate_tiled_raw(s, ell, ops.AGG_POST_DIV, x, None, norm.den, None)
    if getattr(s, "cached_subjects", False):
        return _agg_narrow_tiled(s, ell, norm, x)
    return ops.aggregate_raw(s.rowptr_dst, s.col_dst, norm.w_dst, None, norm.den, None, x, band=band)


def sync_group_of(model):
    """The process group of the model's SyncBatchNorm layers when full-batch statistics across
    ranks are in effect (training, world size > 1), else None."""
    import torch.distributed as dist
    group = None
    for bn in model.batch_norms:
        if isinstance(bn, torch.nn.SyncBatchNorm) and model.training and dist.is_initialized() \
                and dist.get_world_size(bn.process_group) > 1:
            group = bn.process_group if bn.process_group is not None else dist.group.WORLD
    return group


def bn_modules_ok(model) -> bool:
    """Plain affine BatchNorm1d / SyncBatchNorm with running statistics and a fixed momentum."""
    for bn in model.batch_norms:
        if type(bn) not in (torch.nn.BatchNorm1d, torch.nn.SyncBatchNorm) \
                or not (bn.affine and bn.track_running_stats) or bn.momentum is None:
            return False
    return True


def bn_forward_coef(lib, slab, srows, hid, n_nodes, training, gamma, beta, bn, sync_group, sp, dev):
    """BatchNorm coefficient block [a | b | mean | invstd] from the statistics slab (running stats in
    eval mode).  With a sync group the per-rank sums and row count are all-reduced first (one fp64
    block of 2H+1 words; the count never returns to the host).  Returns (coef, count_block)."""
    import torch.distributed as dist
    coef = _f32(dev, 4 * hid)
    count_dev, block = None, None
    if training and sync_group is not None:
        block = torch.empty(2 * hid + 1, dtype=torch.float64, device=dev)
        torch.sum(slab[:srows], dim=0, out=block[:2 * hid])
        block[2 * hid:] = float(n_nodes)
        dist.all_reduce(block, op=dist.ReduceOp.SUM, group=sync_group)
        slab, srows, count_dev = block, 1, block.data_ptr() + 8 * 2 * hid
    _lib.check(lib.cgnn_bn_act_finalize(
        _lib.ptr(slab), srows, hid, float(max(n_nodes, 1)), count_dev, int(training), _lib.ptr(gamma),
        _lib.ptr(beta), _lib.ptr(bn.running_mean), _lib.ptr(bn.running_var), float(bn.momentum),
        float(bn.eps), _lib.ptr(bn.num_batches_tracked) if training else None, _lib.ptr(coef), sp),
        "cgnn_bn_act_finalize")
    return coef, block


def bn_backward_coefs(lib, slab, rows, hid, n_nodes, training, sync_group, count_block, sp, dev, out=(None, None)):
    """(dgamma, dbeta, bwc = c1|c2) from the backward statistics slab.  With a sync group the sums
    are all-reduced for c1|c2 while dgamma/dbeta stay the rank-local sums (the gradient all-reduce
    averages them), exactly like torch's SyncBatchNorm."""
    import torch.distributed as dist
    direct = sync_group is None or count_block is None      # (out: armed .grad views, ops.grad_destination)
    dgamma = out[0] if (direct and out[0] is not None) else _f32(dev, hid)
    dbeta = out[1] if (direct and out[1] is not None) else _f32(dev, hid)
    bwc = _f32(dev, 2 * hid)
    if sync_group is None or count_block is None:
        _lib.check(lib.cgnn_bn_act_bwd_finalize(_lib.ptr(slab), rows, hid, float(max(n_nodes, 1)), None,
                                                int(not training), _lib.ptr(dgamma), _lib.ptr(dbeta),
                                                _lib.ptr(bwc), sp), "cgnn_bn_act_bwd_finalize")
        return dgamma, dbeta, bwc
    sums = torch.sum(slab[:rows], dim=0)                       # fp64 [2H] = sum dZ | sum dZ*xhat
    local_dbeta, local_dgamma = sums[:hid].float(), sums[hid:].float()
    dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=sync_group)
    _lib.check(lib.cgnn_bn_act_bwd_finalize(_lib.ptr(sums), 1, hid, 0.0,
                                            count_block.data_ptr() + 8 * 2 * hid, int(not training),
                                            _lib.ptr(dgamma), _lib.ptr(dbeta), _lib.ptr(bwc), sp),
               "cgnn_bn_act_bwd_finalize")
    return local_dgamma, local_dbeta, bwc


def pooled_bn_backward_coefs(lib, dP, fsum, s, hid, n_nodes, training, sp, dev, out=(None, None)):
    """(dgamma, dbeta, bwc) of the LAST layer from the factor sums its pooled forward pass left
    (cgnn_bn_act_pool_fwd's Fsum): the readout's gradient is constant per graph, so no pass over the
    layer's [Nn, H] output is needed for the BatchNorm-backward sums."""
    dgamma = out[0] if out[0] is not None else _f32(dev, hid)
    dbeta = out[1] if out[1] is not None else _f32(dev, hid)
    bwc = _f32(dev, 2 * hid)
    _lib.check(lib.cgnn_bn_act_pool_bwd_finalize(_lib.ptr(dP), _lib.ptr(fsum), _lib.ptr(s.gptr), s.num_graphs, hid,
                                                 float(max(n_nodes, 1)), int(not training), _lib.ptr(dgamma),
                                                 _lib.ptr(dbeta), _lib.ptr(bwc), sp),
               "cgnn_bn_act_pool_bwd_finalize")
    return dgamma, dbeta, bwc


def _linear_fwd_stats(lib, x1, x2, w, b, grid, relu: bool = True):
    """act([x1 | x2] W^T + b) and the per-workgroup (sum | sum of squares) slab of the result, or
    (None, None) when the shape is outside the weight-stationary kernel."""
    m, k1 = x1.shape
    k2 = 0 if x2 is None else x2.shape[1]
    n = w.shape[0]
    y = torch.empty(m, n, dtype=torch.float32, device=x1.device)
    slab = torch.empty(grid, 2 * n, dtype=torch.float64, device=x1.device)
    rc = lib.cgnn_linear_fwd_stats_f32(
        _lib.ptr(x1), x1.stride(0), k1, _lib.ptr(x2), 0 if x2 is None else x2.stride(0), k2,
        _lib.ptr(w), _lib.ptr(b), int(relu), _lib.ptr(y), y.stride(0), m, n, _lib.ptr(slab), _lib.nbytes(slab),
        _lib.stream_ptr())
    if rc == _lib.CGNN_EUNSUPPORTED:
        return None, None
    _lib.check(rc, "cgnn_linear_fwd_stats_f32")
    return y, slab


class SageEncode(torch.autograd.Function):
    """P[B,H] = mean-pool(SAGE stack(x0)); args = x0, cfg, then (W, b, gamma, beta) per layer."""

    @staticmethod
    def forward(ctx, x0, cfg, *params):
        lib = _lib.load()
        s: BatchStructure = cfg["structure"]
        bns_mod = cfg["batch_norms"]
        training: bool = cfg["training"]
        p: float = cfg["dropout"] if training else 0.0
        rng = cfg.get("rng_state")
        L = len(params) // 4
        dev = x0.device
        _sp = _lib.stream_ptr(dev)          # one lookup per pass (torch.cuda.current_stream is ~10 us)
        st = lambda: _sp
        x = x0.contiguous()
        n_nodes = s.num_nodes
        sv = _Saved()
        sv.s, sv.p, sv.training = s, p, training
        grid = int(lib.cgnn_fused_grid())
        # graphs of <= 384 nodes: LDS tiles over the blocked-ELL; larger ones: the CSR gather kernel, its dense
        # fragments on the matrix cores where the batch has them (the transposed pass then needs w / den per edge)
        sv.tiled = s.tiled_ok(params[0].shape[0])
        sv.ell = s.fused_meta(TILE_ROWS, grid, 0.0) if sv.tiled else None
        sv.norm = s.sage_norm(backward_coef=not sv.tiled)
        sv.band = (None, None) if sv.tiled else s.band_ops("sage", sv.norm)
        sv.xs, sv.aggs, sv.zs, sv.coefs, sv.masks, sv.ws = [], [], [], [], [], []
        sv.xa0 = None
        sv.sync_group, sv.count_block = cfg.get("sync_group"), None
        sv.grad_dst = cfg.get("grad_dst") or [None] * len(params)
        with _lib.device_guard(dev):
            if rng is not None and p > 0:
                _lib.check(lib.cgnn_rng_advance(_lib.ptr(rng), L + 1, st()), "cgnn_rng_advance")
            rows = int(lib.cgnn_bn_act_slab_rows(n_nodes))
            pending = None        # (z, coef, seed, rw, mask) of the previous layer: its BatchNorm+dropout
                                  # is applied by the consumer below, not by a pass of its own
            for li in range(L):
                w, b, gamma, beta = (t.contiguous() for t in params[4 * li:4 * li + 4])
                hid = w.shape[0]
                agg = None
                if pending is not None:
                    pz, pcoef, pseed, prw, pmask = pending
                    x = torch.empty_like(pz)
                    if sv.tiled and s.tiled_ok(pz.shape[1]):
                        # X' = drop(BatchNorm(Z)) formed while the aggregate stages its tiles (and
                        # written out for the projection): no apply pass
                        agg = ops.aggregate_tiled_bn_raw(s, sv.ell, ops.AGG_POST_DIV, pz, None, sv.norm.den, None,
                                                         pcoef, False, p, pseed, prw, pmask, x)
                    else:
                        _lib.check(lib.cgnn_bn_act_fwd_apply(_lib.ptr(pz), _lib.ptr(pcoef), 0, p, pseed, prw,
                                                             _lib.ptr(pmask), _lib.ptr(x), n_nodes, pz.shape[1], st()),
                                   "cgnn_bn_act_fwd_apply")
                    pending = None
                fin = x.shape[1]
                slab = None
                srows = rows
                if li == 0 and 2 * fin <= PAD_K and hid in (64, 128, 256) and n_nodes >= PAD_MIN_ROWS:
                    # narrow input layer: pack [x0 | agg(x0) | 0] and the zero-padded weight into
                    # 32-wide panels so that the tall weight-stationary GEMMs apply (K = 10 would
                    # otherwise run the per-tile kernels at a few % of the matrix-core rate)
                    xa = torch.nn.functional.pad(x, (0, PAD_K - fin))        # one pass: [x0 | 0]
                    if getattr(s, "cached_subjects", False):
                        agg = _agg_narrow_tiled(s, sv.ell, sv.norm, x, out=xa[:, fin:2 * fin])
                    else:
                        agg = ops.aggregate_raw(s.rowptr_dst, s.col_dst, sv.norm.w_dst, None, sv.norm.den,
                                                None, x, out=xa[:, fin:2 * fin])
                    wp = torch.nn.functional.pad(w, (0, PAD_K - 2 * fin))
                    sv.xa0 = xa
                    gemm_in = (xa, None, wp)
                else:
                    if agg is None:
                        agg = _agg_fwd(s, sv.ell, sv.norm, x, sv.band[0])
                    gemm_in = (x, agg, w)
                z = None
                if training:
                    # projection with the BatchNorm statistics of its output in the epilogue
                    z, slab = _linear_fwd_stats(lib, *gemm_in, b, grid)
                    srows = grid
                if z is None:
                    z = ops.linear_fwd_raw(*gemm_in, b, True)
                    if training:
                        slab = torch.empty(rows, 2 * hid, dtype=torch.float64, device=dev)
                        srows = rows
                        _lib.check(lib.cgnn_bn_act_fwd_stats(_lib.ptr(z), n_nodes, hid, _lib.ptr(slab), _lib.nbytes(slab), st()),
                                   "cgnn_bn_act_fwd_stats")
                coef, blk = bn_forward_coef(lib, slab, srows, hid, n_nodes, training, gamma, beta,
                                            bns_mod[li], sv.sync_group, st(), dev)
                sv.count_block = blk if blk is not None else sv.count_block
                mask = torch.empty(n_nodes * (hid // 4), dtype=torch.uint8, device=dev) if p > 0 else None
                seed = _lib.next_seed(dev) if p > 0 else 0
                rw = None if (rng is None or p <= 0) else rng.data_ptr() + 4 * li
                sv.xs.append(x); sv.aggs.append(agg); sv.zs.append(z); sv.coefs.append(coef)
                sv.masks.append(mask); sv.ws.append(w)
                if li == L - 1:
                    # last layer: BatchNorm + dropout + mean-pool in one pass, X' never written
                    pooled = _f32(dev, s.num_graphs, hid)
                    # (factor sums for the backward statistics; under sync-BN the sums are exchanged
                    # through the slab of the ordinary statistics pass instead)
                    sv.fsum = _f32(dev, 2, s.num_graphs, hid) if (sv.sync_group is None and any(ctx.needs_input_grad)) else None
                    _lib.check(lib.cgnn_bn_act_pool_fwd(_lib.ptr(z), _lib.ptr(coef), 0, p, seed, rw,
                                                        _lib.ptr(mask), _lib.ptr(s.gptr), s.num_graphs,
                                                        _lib.ptr(pooled), hid, _lib.ptr(sv.fsum), st()),
                               "cgnn_bn_act_pool_fwd")
                    break
                pending = (z, coef, seed, rw, mask)
        if cfg.get("record") is not None:
            cfg["record"]["layers"] = list(sv.masks)
        ctx.sv = sv
        ctx.L = L
        return pooled

    @staticmethod
    def backward(ctx, dP):
        lib = _lib.load()
        sv: _Saved = ctx.sv
        s, L = sv.s, ctx.L
        dev = dP.device
        _sp = _lib.stream_ptr(dev)          # one lookup per pass (torch.cuda.current_stream is ~10 us)
        st = lambda: _sp
        n_nodes = s.num_nodes
        dP = dP.contiguous()
        grads: List[Optional[torch.Tensor]] = [None] * (4 * L)
        dst = sv.grad_dst
        with _lib.device_guard(dev):
            dx = None                      # last layer: gradient rebuilt from dP inside the kernels
            deferred = _lib.DeferredReduce()
            rows = int(lib.cgnn_bn_act_slab_rows(n_nodes))
            for li in range(L - 1, -1, -1):
                x, agg, z, coef, mask, w = (sv.xs[li], sv.aggs[li], sv.zs[li], sv.coefs[li],
                                            sv.masks[li], sv.ws[li])
                hid, fin = w.shape[0], x.shape[1]
                bn_out = (dst[4 * li + 2], dst[4 * li + 3])
                # ---- BatchNorm + dropout backward, ReLU' of the layer and db in two passes
                pool = (_lib.ptr(dP), _lib.ptr(s.node_graph), _lib.ptr(s.gptr)) if li == L - 1 \
                    else (None, None, None)
                if li == L - 1 and sv.fsum is not None:
                    dgamma, dbeta, bwc = pooled_bn_backward_coefs(lib, dP, sv.fsum, s, hid, n_nodes, sv.training, st(), dev,
                                                                  bn_out)
                else:
                    slab = torch.empty(rows, 2 * hid, dtype=torch.float64, device=dev)
                    _lib.check(lib.cgnn_bn_act_bwd_stats(_lib.ptr(dx), _lib.ptr(z), _lib.ptr(mask),
                                                         _lib.ptr(coef), 0, sv.p, n_nodes, hid,
                                                         _lib.ptr(slab), _lib.nbytes(slab), *pool, st()), "cgnn_bn_act_bwd_stats")
                    dgamma, dbeta, bwc = bn_backward_coefs(lib, slab, rows, hid, n_nodes, sv.training,
                                                           sv.sync_group, sv.count_block, st(), dev, bn_out)
                cs_rows = int(lib.cgnn_bn_act_apply_blocks(n_nodes, hid))
                cs_slab = torch.empty(cs_rows, hid, dtype=torch.float64, device=dev)
                dpre = torch.empty_like(z)
                _lib.check(lib.cgnn_bn_act_bwd_apply(_lib.ptr(dx), _lib.ptr(z), _lib.ptr(mask),
                                                     _lib.ptr(coef), _lib.ptr(bwc), 0, sv.p, 1,
                                                     _lib.ptr(cs_slab), _lib.nbytes(cs_slab), _lib.ptr(dpre), n_nodes, hid,
                                                     *pool, st()), "cgnn_bn_act_bwd_apply")
                db = dst[4 * li + 1] if dst[4 * li + 1] is not None else _f32(dev, hid)
                deferred.add(cs_slab, cs_rows, hid, db)       # all layers' db: one launch at the end
                # ---- dW = dPre^T [X | A]
                if li == 0 and sv.xa0 is not None:
                    dwp = _f32(dev, hid, PAD_K)
                    ops.linear_bwd_weight_raw(dpre, sv.xa0, dwp, 0)
                    grads[0:4] = [dwp[:, :2 * fin].contiguous(), db, dgamma, dbeta]
                    break
                dw = dst[4 * li] if dst[4 * li] is not None else torch.empty_like(w)
                ops.linear_bwd_weight2_raw(dpre, x, agg, dw)
                grads[4 * li:4 * li + 4] = [dw, db, dgamma, dbeta]
                if li == 0:
                    break
                # ---- [dX1 | dA] = dPre W, then dX = dX1 + A_w^T (dA / den)
                dcat = ops.linear_bwd_input_raw(dpre, w, 0, 2 * fin)
                if sv.tiled:
                    dx = ops.aggregate_tiled_raw(s, sv.ell, ops.AGG_TRANSPOSED | ops.AGG_PRE_DIV,
                                                 dcat[:, fin:], sv.norm.den, None, None, yadd=dcat[:, :fin])
                else:
                    dx = ops.aggregate_raw(s.rowptr_src, s.col_src, sv.norm.coef_src_bwd, None, None, None,
                                           dcat[:, fin:], band=sv.band[1], yadd=dcat[:, :fin])
            deferred.flush(st())
        ctx.sv = None
        return (None, None, *ops.undelivered(grads, dst))


def encode(model, batch, structure: BatchStructure) -> torch.Tensor:
    params = []
    for conv, bn in zip(model.convs, model.batch_norms):
        params += [conv.linear.weight, conv.linear.bias, bn.weight, bn.bias]
    from .structure import twin_view, unpermute_record
    structure, x0, twin = twin_view(structure, batch.node_features)
    cfg = {"structure": structure, "batch_norms": list(model.batch_norms), "training": model.training,
           "dropout": float(model.dropout), "rng_state": getattr(model, "rng_device_state", None),
           "sync_group": sync_group_of(model), "record": model._dropout_record(),
           "grad_dst": ops.claim_destinations(params, model.training)}
    out = SageEncode.apply(x0, cfg, *params)
    unpermute_record(twin, cfg.get("record"))
    return out
