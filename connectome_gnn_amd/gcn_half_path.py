"""GCN encoder with fp16-STORAGE activations for large dense parcellations (BASELINE config 5:
1000-ROI graphs at 10 % density, hidden 256) -- one autograd node, host orchestration only.

The reference has no fp16 path (models.py:46,97,103,147 hard-code fp32; ``.half()`` raises), so
this is new: every [Nn, H] array that crosses HBM is IEEE half, every accumulation is fp32 (fp64 for
the BatchNorm statistics), parameters and their gradients stay fp32.  Selected with
``GCNConnectome(..., storage="fp16")``; validated against the fp32 oracle at fp16 resolution.

At ~100 neighbours per node the per-edge forms of the aggregation are bound by vector/LDS work
per edge, so the operator is applied DENSE, per graph, on the fp16 matrix cores:

  once per batch    Mf, Mb = dense D^-1/2 (A + I) D^-1/2 of every graph and its transpose, half,
                    MFMA-fragment-major                         cgnn_dense_adj_f16 (static, cached)
  layer 0           P0 = Mf X0 on a 64-column half panel (narrow: a quarter of a 256-wide pass),
                    Y0 = P0 W0^T + b0
  layer l > 0       T = X W^T                       cgnn_linear_fwd_f16 (weight-stationary, fp32 accumulate)
                    Y = Mf T + b                    cgnn_dense_aggregate_f16 (v_mfma_f32_32x32x16_f16)
  every layer       X' = dropout(relu(BatchNorm(Y)))            cgnn_bn_act_*_f16 (two passes)
  readout           fused into the last BatchNorm pass          cgnn_bn_act_pool_fwd_f16
  backward          dY = BatchNorm'(...) (two passes, db = column sums), dT = Mb dY,
                    dW = dT^T X (cgnn_linear_bwd_weight_f16: LDS transposing reads, fp32 partials
                    per run of rows), dX = dT W (cgnn_linear_bwd_input_f16);
                    layer 0: dW0 = dY0^T P0, no aggregation

Every product is a hand-written kernel of csrc/gemm_h16.hip (v_mfma_f32_32x32x16_f16, weights
converted fp32 -> half inside the kernels): nothing on this path goes to a GEMM library.
"""
from __future__ import annotations

from typing import List, Optional

import torch

from . import _lib, ops
from .sage_path import bn_modules_ok, pooled_bn_backward_coefs
from .structure import BatchStructure

MAX_NODES = 1024          # dense pitch limit of cgnn_dense_adj_f16


def eligible(model, batch, structure: BatchStructure) -> Optional[str]:
    hid = model.convs[0].linear.weight.shape[0]
    if hid not in (64, 128, 256):
        return "hidden_dim is not 64, 128 or 256 (the half projections of gemm_h16.hip)"
    if not structure.block_diagonal:
        return "edges cross graph boundaries"
    if structure.max_nodes_per_graph > MAX_NODES:
        return f"a graph has more than {MAX_NODES} nodes"
    if batch.node_features.requires_grad:
        return "node_features require grad"
    if model.convs[0].linear.weight.shape[1] > P0_COLS:
        return f"more than {P0_COLS} input features"
    if not bn_modules_ok(model) or any(isinstance(bn, torch.nn.SyncBatchNorm) for bn in model.batch_norms):
        return "BatchNorm is not a plain affine BatchNorm1d with running stats"
    return None


def _agg(s, m, x, bias=None, stat_slab=None):
    if isinstance(m, ops.DensePack):
        return ops.dense_aggregate_c16_raw(s, m, x, bias, stat_slab)
    return ops.dense_aggregate_f16_raw(s, m, x, bias, stat_slab)


PACK_BELOW = 0.6       # use the per-fragment operator when it is at most this fraction of the dense bytes


def _operator(s: BatchStructure, coef, selfc, transposed: bool):
    """The dense operator of one ordering in the smaller of its two forms: per MFMA fragment
    (ops.DensePack: nearly full fragments dense, the others as entry lists) when that is clearly
    smaller -- the aggregate is bound by the operator's bytes, which come from HBM on each of
    the five launches of a step -- else the plain dense [B, P, P] half matrix."""
    pk = ops.dense_pack_f16(s, coef, selfc, transposed)
    pitch = pk.pitch
    if pk.nbytes() <= PACK_BELOW * (2.0 * s.num_graphs * pitch * pitch):
        return pk
    return ops.dense_adj_f16(s, coef, selfc, transposed)


def dense_operators(s: BatchStructure):
    """(Mf, Mb): normalised operator and its transpose, built once per batch structure."""
    cached = s.__dict__.get("_dense_f16")
    if cached is None:
        norm = s.gcn_norm()
        cached = (_operator(s, norm.coef_dst, norm.selfc, False), _operator(s, norm.coef_src, norm.selfc, True))
        s.__dict__["_dense_f16"] = cached
    return cached


P0_COLS = 64             # layer 0's input features ride in one 64-column half panel


class _Saved:
    __slots__ = ("s", "mb", "xs", "ys", "coefs", "masks", "ws", "p0", "p", "training", "fsum")


def _f32(dev, *shape):
    return torch.empty(*shape, dtype=torch.float32, device=dev)


class GcnHalfEncode(torch.autograd.Function):
    """P[B,H] = mean-pool(GCN stack(x0)), activations stored as half."""

    @staticmethod
    def forward(ctx, x0, cfg, *params):
        lib = _lib.load()
        s: BatchStructure = cfg["structure"]
        bns_mod = cfg["batch_norms"]
        training: bool = cfg["training"]
        p: float = cfg["dropout"] if training else 0.0
        L = len(params) // 4
        dev = x0.device
        sp = _lib.stream_ptr(dev)
        n_nodes, B = s.num_nodes, s.num_graphs
        mf, mb = dense_operators(s)
        sv = _Saved()
        sv.s, sv.mb, sv.p, sv.training = s, mb, p, training
        sv.xs, sv.ys, sv.coefs, sv.masks, sv.ws = [], [], [], [], []
        x = None
        rng = cfg.get("rng_state")          # device words a captured step refreshes per replay
        with _lib.device_guard(dev):
            rows = int(lib.cgnn_bn_act_slab_rows(n_nodes))
            if rng is not None and p > 0:
                _lib.check(lib.cgnn_rng_advance(_lib.ptr(rng), L + 1, sp), "cgnn_rng_advance")
            for li in range(L):
                w, b, gamma, beta = (t.contiguous() for t in params[4 * li:4 * li + 4])
                hid = w.shape[0]
                if li == 0:
                    # A_hat (X0 W0^T) == (A_hat X0) W0^T: aggregate the few input columns (one
                    # 64-column half panel through the dense operator), then project
                    sv.p0 = _agg(s, mf, ops.pad_cast_f16(x0, P0_COLS))    # [Nn, 64] half, cols >= F0 zero
                    y = None
                    if training:                                           # statistics in the epilogue
                        y, slab = ops.linear_fwd_stats_f16_raw(sv.p0, w, b, int(lib.cgnn_fused_grid()))
                        srows = int(lib.cgnn_fused_grid())
                    if y is None:
                        y = ops.linear_fwd_f16_raw(sv.p0, w, b)           # K = 64 panel, W0 [H, F0]
                        slab, srows = None, rows
                else:
                    slab, srows = None, rows
                if li > 0:
                    t = ops.linear_fwd_f16_raw(x, w)                       # half in / out, fp32 accumulate
                    if training:                                           # statistics in the epilogue
                        srows = int(lib.cgnn_fused_grid())
                        slab = torch.empty(srows, 2 * hid, dtype=torch.float64, device=dev)
                    y = _agg(s, mf, t, b, slab)
                if training and slab is None:
                    slab = torch.empty(rows, 2 * hid, dtype=torch.float64, device=dev)
                    _lib.check(lib.cgnn_bn_act_fwd_stats_f16(_lib.ptr(y), n_nodes, hid, _lib.ptr(slab), _lib.nbytes(slab), sp),
                               "cgnn_bn_act_fwd_stats_f16")
                bn = bns_mod[li]
                coef = _f32(dev, 4 * hid)
                _lib.check(lib.cgnn_bn_act_finalize(
                    _lib.ptr(slab), srows, hid, float(max(n_nodes, 1)), None, int(training), _lib.ptr(gamma),
                    _lib.ptr(beta), _lib.ptr(bn.running_mean), _lib.ptr(bn.running_var), float(bn.momentum),
                    float(bn.eps), _lib.ptr(bn.num_batches_tracked) if training else None, _lib.ptr(coef), sp),
                    "cgnn_bn_act_finalize")
                mask = torch.empty(n_nodes * (hid // 4), dtype=torch.uint8, device=dev) if p > 0 else None
                seed = _lib.next_seed(dev) if p > 0 else 0
                rw = None if (rng is None or p <= 0) else rng.data_ptr() + 4 * li
                sv.xs.append(x); sv.ys.append(y); sv.coefs.append(coef); sv.masks.append(mask); sv.ws.append(w)
                if li == L - 1:
                    pooled = _f32(dev, B, hid)
                    # per-graph factor sums: the backward statistics of this layer need no pass over Y
                    sv.fsum = _f32(dev, 2, B, hid) if any(ctx.needs_input_grad) else None
                    _lib.check(lib.cgnn_bn_act_pool_fwd_f16(_lib.ptr(y), _lib.ptr(coef), 1, p, seed, rw,
                                                            _lib.ptr(mask), _lib.ptr(s.gptr), B,
                                                            _lib.ptr(pooled), hid, _lib.ptr(sv.fsum), sp),
                               "cgnn_bn_act_pool_fwd_f16")
                    break
                xn = torch.empty_like(y)
                _lib.check(lib.cgnn_bn_act_fwd_apply_f16(_lib.ptr(y), _lib.ptr(coef), 1, p, seed, rw,
                                                         _lib.ptr(mask), _lib.ptr(xn), n_nodes, hid, sp),
                           "cgnn_bn_act_fwd_apply_f16")
                x = xn
        if cfg.get("record") is not None:
            cfg["record"]["layers"] = list(sv.masks)
        ctx.sv, ctx.L = sv, L
        return pooled

    @staticmethod
    def backward(ctx, dP):
        lib = _lib.load()
        sv: _Saved = ctx.sv
        s, L = sv.s, ctx.L
        dev = dP.device
        sp = _lib.stream_ptr(dev)
        n_nodes = s.num_nodes
        dP = dP.contiguous()
        grads: List[Optional[torch.Tensor]] = [None] * (4 * L)
        with _lib.device_guard(dev):
            rows = int(lib.cgnn_bn_act_slab_rows(n_nodes))
            dx = None                           # last layer: gradient rebuilt from dP inside the kernels
            deferred = _lib.DeferredReduce()
            for li in range(L - 1, -1, -1):
                x, y, coef, mask, w = sv.xs[li], sv.ys[li], sv.coefs[li], sv.masks[li], sv.ws[li]
                hid = w.shape[0]
                pool = (_lib.ptr(dP), _lib.ptr(s.node_graph), _lib.ptr(s.gptr)) if li == L - 1 else (None, None, None)
                if li == L - 1 and sv.fsum is not None:
                    dgamma, dbeta, bwc = pooled_bn_backward_coefs(lib, dP, sv.fsum, s, hid, n_nodes, sv.training, sp, dev)
                else:
                    slab, srows = torch.empty(rows, 2 * hid, dtype=torch.float64, device=dev), rows
                    _lib.check(lib.cgnn_bn_act_bwd_stats_f16(_lib.ptr(dx), _lib.ptr(y), _lib.ptr(mask), _lib.ptr(coef),
                                                             1, sv.p, n_nodes, hid, _lib.ptr(slab), _lib.nbytes(slab), *pool, sp),
                               "cgnn_bn_act_bwd_stats_f16")
                    dgamma, dbeta, bwc = _f32(dev, hid), _f32(dev, hid), _f32(dev, 2 * hid)
                    _lib.check(lib.cgnn_bn_act_bwd_finalize(_lib.ptr(slab), srows, hid, float(max(n_nodes, 1)), None,
                                                            int(not sv.training), _lib.ptr(dgamma), _lib.ptr(dbeta),
                                                            _lib.ptr(bwc), sp), "cgnn_bn_act_bwd_finalize")
                db = _f32(dev, hid)
                if li > 0 and isinstance(sv.mb, ops.DensePack):
                    # dT = A_hat^T dY with dY formed while the aggregate stages its slices: no apply
                    # pass, dY is never written (db from the per-graph column sums it leaves)
                    dt, cs_slab = ops.dense_aggregate_c16_bnbwd_raw(
                        s, sv.mb, dx, dP if li == L - 1 else None, y, mask, coef, bwc, True, sv.p)
                    deferred.add(cs_slab, s.num_graphs, hid, db)
                    dy = None
                else:
                    cs_rows = int(lib.cgnn_bn_act_apply_blocks(n_nodes, hid))
                    cs_slab = torch.empty(cs_rows, hid, dtype=torch.float64, device=dev)
                    dy = torch.empty_like(y)
                    _lib.check(lib.cgnn_bn_act_bwd_apply_f16(_lib.ptr(dx), _lib.ptr(y), _lib.ptr(mask), _lib.ptr(coef),
                                                             _lib.ptr(bwc), 1, sv.p, 0, _lib.ptr(cs_slab), _lib.nbytes(cs_slab), _lib.ptr(dy),
                                                             n_nodes, hid, *pool, sp), "cgnn_bn_act_bwd_apply_f16")
                    deferred.add(cs_slab, cs_rows, hid, db)       # all layers' db: one launch at the end
                    dt = None
                if li == 0:
                    # Y0 = (A_hat X0) W0^T + b0: dW0 = dY0^T P0, no aggregation in the backward
                    dw = ops.linear_bwd_weight_f16_raw(dy, sv.p0, w.shape[1])
                    grads[0:4] = [dw, db, dgamma, dbeta]
                    break
                if dt is None:
                    dt = _agg(s, sv.mb, dy)         # dT = A_hat^T dY
                grads[4 * li:4 * li + 4] = [ops.linear_bwd_weight_f16_raw(dt, x), db, dgamma, dbeta]
                dx = ops.linear_bwd_input_f16_raw(dt, w)                   # dX = dT W
            deferred.flush(sp)
        ctx.sv = None
        return (None, None, *grads)


def encode(model, batch, structure: BatchStructure) -> torch.Tensor:
    params = []
    for conv, bn in zip(model.convs, model.batch_norms):
        params += [conv.linear.weight, conv.bias, bn.weight, bn.bias]
    cfg = {"structure": structure, "batch_norms": list(model.batch_norms), "training": model.training,
           "dropout": float(model.dropout), "record": model._dropout_record(),
           "rng_state": getattr(model, "rng_device_state", None)}
    return GcnHalfEncode.apply(batch.node_features, cfg, *params)
