"""GCN encoder as one autograd node for hidden = 128, 256, ... (hidden 64 has the per-tile fused
kernels of fused.py): host orchestration, the GCN counterpart of sage_path.py.

GCNConnectome.encode (reference models.py:203-211) with hand-written backward, so that every pass
over a node array is one HIP kernel doing several things at once:

  forward, layer 0     P0 = A_hat X0 (narrow)                   cgnn_aggregate_f32
                       Y0 = P0 W0^T + b (+ BatchNorm statistics) cgnn_linear_fwd_stats_f32 (packed K=32)
  forward, layer l>0   T  = X W^T                                cgnn_linear_fwd_f32 (W in LDS)
                       Y  = dis * (A_w + I)(dis * T) + b         cgnn_aggregate_tiled_f32 (LDS tiles); graphs of
                                                                 more than 384 nodes: cgnn_aggregate_f32 (CSR
                                                                 gather) + cgnn_band_aggregate_f32 (dense
                                                                 fragments as split-bf16 MFMA products)
  every layer          X' = dropout(relu(BatchNorm(Y)))          cgnn_bn_act_* ; last layer fused with
                                                                 the readout (cgnn_bn_act_pool_fwd)
  backward, layer l    dY = BatchNorm'(dX' * drop' * relu'), db = colsum(dY)   cgnn_bn_act_bwd_*
                       dT = dis * (A_w + I)^T (dis * dY)         cgnn_aggregate_tiled_f32 (transposed)
                       dW = dT^T X ; dX = dT W                   cgnn_linear_bwd_weight/input_f32
  backward, layer 0    dW0 = dY0^T P0                            cgnn_linear_bwd_weight_f32 (packed)

No arithmetic of the path happens in torch here; torch allocates buffers and orders the launches.
"""
from __future__ import annotations

from typing import List, Optional

import torch

from . import _lib, ops
from .sage_path import (PAD_K, PAD_MIN_ROWS, TILE_ROWS, _f32, _linear_fwd_stats, bn_backward_coefs,
                        pooled_bn_backward_coefs,
                        bn_forward_coef, bn_modules_ok, sync_group_of)
from .structure import BatchStructure


def eligible(model, batch, structure: BatchStructure) -> Optional[str]:
    """None if this path covers (model, batch); else the reason it does not."""
    hid = model.convs[0].linear.weight.shape[0]
    if hid % 64 or not bool(_lib.load().cgnn_bn_act_width_ok(hid)):
        return "hidden_dim is not 64, 128, 256, ..."
    if model.convs[0].linear.weight.shape[1] >= hid:
        return "input features are not narrower than hidden_dim"
    if not structure.tiled_ok(hid) and not isinstance(structure, BatchStructure):
        return "graphs do not fit an LDS tile and the structure has no CSR form"
    if batch.node_features.requires_grad:
        return "node_features require grad"
    if not bn_modules_ok(model):
        return "BatchNorm is not a plain affine BatchNorm1d / SyncBatchNorm with running stats"
    return None


class _Saved:
    __slots__ = ("s", "ell", "norm", "xs", "ys", "coefs", "masks", "p", "training", "ws", "p0", "padded",
                 "sync_group", "count_block", "fsum", "tiled", "band", "grad_dst")


class GcnWideEncode(torch.autograd.Function):
    """P[B,H] = mean-pool(GCN stack(x0)); args = x0, cfg, then (W, b, gamma, beta) per layer."""

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
        grid = int(lib.cgnn_fused_grid())
        sv = _Saved()
        sv.s, sv.p, sv.training = s, p, training
        sv.norm = s.gcn_norm()
        hid_all = params[0].shape[0]
        sv.tiled = s.tiled_ok(hid_all)
        # graphs of <= 384 nodes: LDS tiles over the blocked-ELL (with the self-loop entry); larger ones:
        # the CSR gather kernel, its dense fragments on the matrix cores where the batch has them
        sv.ell = s.fused_meta(TILE_ROWS, grid, 1.0) if sv.tiled else None
        sv.band = (None, None) if sv.tiled else s.band_ops("gcn", sv.norm)
        sv.xs, sv.ys, sv.coefs, sv.masks, sv.ws = [], [], [], [], []
        sv.p0, sv.padded = None, False
        sv.sync_group, sv.count_block = cfg.get("sync_group"), None
        sv.grad_dst = cfg.get("grad_dst") or [None] * len(params)
        nrm = sv.norm
        with _lib.device_guard(dev):
            if rng is not None and p > 0:
                _lib.check(lib.cgnn_rng_advance(_lib.ptr(rng), L + 1, st()), "cgnn_rng_advance")
            rows = int(lib.cgnn_bn_act_slab_rows(n_nodes))
            for li in range(L):
                w, b, gamma, beta = (t.contiguous() for t in params[4 * li:4 * li + 4])
                hid, fin = w.shape[0], x.shape[1]
                slab, srows, y = None, rows, None
                if li == 0:
                    # narrow input: aggregate first (A_hat (X W^T) == (A_hat X) W^T, models.py:111-114)
                    pad = fin <= PAD_K and hid in (64, 128, 256) and n_nodes >= PAD_MIN_ROWS
                    width = PAD_K if pad else fin
                    p0 = torch.zeros(n_nodes, width, dtype=torch.float32, device=dev) if pad \
                        else _f32(dev, n_nodes, fin)
                    ops.aggregate_raw(s.rowptr_dst, s.col_dst, nrm.coef_dst, nrm.selfc, None, None, x,
                                      out=p0[:, :fin])
                    wq = w
                    if pad:
                        wq = torch.zeros(hid, PAD_K, dtype=torch.float32, device=dev)
                        wq[:, :fin].copy_(w)
                    if training:
                        y, slab = _linear_fwd_stats(lib, p0, None, wq, b, grid, relu=False)
                        srows = grid
                    if y is None:
                        y = ops.linear_fwd_raw(p0, None, wq, b, False)
                    sv.p0, sv.padded = p0, pad
                else:
                    t = ops.linear_fwd_raw(x, None, w, None, False)
                    if sv.tiled:
                        y = ops.aggregate_tiled_raw(s, sv.ell, 0, t, nrm.dis, nrm.dis, b)
                    else:
                        y = ops.aggregate_raw(s.rowptr_dst, s.col_dst, nrm.coef_dst, nrm.selfc, None, b, t,
                                              band=sv.band[0])
                if training and slab is None:
                    slab = torch.empty(rows, 2 * hid, dtype=torch.float64, device=dev)
                    srows = rows
                    _lib.check(lib.cgnn_bn_act_fwd_stats(_lib.ptr(y), n_nodes, hid, _lib.ptr(slab), _lib.nbytes(slab), st()),
                               "cgnn_bn_act_fwd_stats")
                coef, blk = bn_forward_coef(lib, slab, srows, hid, n_nodes, training, gamma, beta,
                                            bns_mod[li], sv.sync_group, st(), dev)
                sv.count_block = blk if blk is not None else sv.count_block
                mask = torch.empty(n_nodes * (hid // 4), dtype=torch.uint8, device=dev) if p > 0 else None
                seed = _lib.next_seed(dev) if p > 0 else 0
                rw = None if (rng is None or p <= 0) else rng.data_ptr() + 4 * li
                sv.xs.append(x); sv.ys.append(y); sv.coefs.append(coef); sv.masks.append(mask); sv.ws.append(w)
                if li == L - 1:
                    pooled = _f32(dev, s.num_graphs, hid)
                    sv.fsum = _f32(dev, 2, s.num_graphs, hid) if (sv.sync_group is None and any(ctx.needs_input_grad)) else None
                    _lib.check(lib.cgnn_bn_act_pool_fwd(_lib.ptr(y), _lib.ptr(coef), 1, p, seed, rw,
                                                        _lib.ptr(mask), _lib.ptr(s.gptr), s.num_graphs,
                                                        _lib.ptr(pooled), hid, _lib.ptr(sv.fsum), st()), "cgnn_bn_act_pool_fwd")
                    break
                xn = torch.empty_like(y)
                _lib.check(lib.cgnn_bn_act_fwd_apply(_lib.ptr(y), _lib.ptr(coef), 1, p, seed, rw,
                                                     _lib.ptr(mask), _lib.ptr(xn), n_nodes, hid, st()),
                           "cgnn_bn_act_fwd_apply")
                x = xn
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
        nrm = sv.norm
        grads: List[Optional[torch.Tensor]] = [None] * (4 * L)
        dst = sv.grad_dst
        with _lib.device_guard(dev):
            dx = None                      # last layer: gradient rebuilt from dP inside the kernels
            deferred = _lib.DeferredReduce()
            rows = int(lib.cgnn_bn_act_slab_rows(n_nodes))
            for li in range(L - 1, -1, -1):
                x, y, coef, mask, w = sv.xs[li], sv.ys[li], sv.coefs[li], sv.masks[li], sv.ws[li]
                hid, fin = w.shape[0], x.shape[1]
                bn_out = (dst[4 * li + 2], dst[4 * li + 3])
                pool = (_lib.ptr(dP), _lib.ptr(s.node_graph), _lib.ptr(s.gptr)) if li == L - 1 \
                    else (None, None, None)
                if li == L - 1 and sv.fsum is not None:
                    dgamma, dbeta, bwc = pooled_bn_backward_coefs(lib, dP, sv.fsum, s, hid, n_nodes, sv.training, st(), dev,
                                                                  bn_out)
                else:
                    slab = torch.empty(rows, 2 * hid, dtype=torch.float64, device=dev)
                    _lib.check(lib.cgnn_bn_act_bwd_stats(_lib.ptr(dx), _lib.ptr(y), _lib.ptr(mask),
                                                         _lib.ptr(coef), 1, sv.p, n_nodes, hid,
                                                         _lib.ptr(slab), _lib.nbytes(slab), *pool, st()), "cgnn_bn_act_bwd_stats")
                    dgamma, dbeta, bwc = bn_backward_coefs(lib, slab, rows, hid, n_nodes, sv.training,
                                                           sv.sync_group, sv.count_block, st(), dev, bn_out)
                cs_rows = int(lib.cgnn_bn_act_apply_blocks(n_nodes, hid))
                cs_slab = torch.empty(cs_rows, hid, dtype=torch.float64, device=dev)
                dy = torch.empty_like(y)
                _lib.check(lib.cgnn_bn_act_bwd_apply(_lib.ptr(dx), _lib.ptr(y), _lib.ptr(mask),
                                                     _lib.ptr(coef), _lib.ptr(bwc), 1, sv.p, 0,
                                                     _lib.ptr(cs_slab), _lib.nbytes(cs_slab), _lib.ptr(dy), n_nodes, hid,
                                                     *pool, st()), "cgnn_bn_act_bwd_apply")
                db = dst[4 * li + 1] if dst[4 * li + 1] is not None else _f32(dev, hid)
                deferred.add(cs_slab, cs_rows, hid, db)       # all layers' db: one launch at the end
                if li == 0:
                    # Y0 = P0 W0^T + b  ->  dW0 = dY0^T P0 (no aggregation in the backward)
                    if sv.padded:
                        dwp = _f32(dev, hid, PAD_K)
                        ops.linear_bwd_weight_raw(dy, sv.p0, dwp, 0)
                        dw = dwp[:, :fin].contiguous()
                    else:
                        dw = dst[0] if dst[0] is not None else torch.empty_like(w)
                        ops.linear_bwd_weight_raw(dy, sv.p0, dw, 0)
                    grads[0:4] = [dw, db, dgamma, dbeta]
                    break
                if sv.tiled:
                    dt = ops.aggregate_tiled_raw(s, sv.ell, ops.AGG_TRANSPOSED, dy, nrm.dis, nrm.dis, None)
                else:
                    dt = ops.aggregate_raw(s.rowptr_src, s.col_src, nrm.coef_src, nrm.selfc, None, None, dy,
                                           band=sv.band[1])
                dw = dst[4 * li] if dst[4 * li] is not None else torch.empty_like(w)
                ops.linear_bwd_weight_raw(dt, x, dw, 0)
                grads[4 * li:4 * li + 4] = [dw, db, dgamma, dbeta]
                dx = ops.linear_bwd_input_raw(dt, w, 0, fin)
            deferred.flush(st())
        ctx.sv = None
        return (None, None, *ops.undelivered(grads, dst))


def encode(model, batch, structure: BatchStructure) -> torch.Tensor:
    params = []
    for conv, bn in zip(model.convs, model.batch_norms):
        params += [conv.linear.weight, conv.bias, bn.weight, bn.bias]
    from .structure import twin_view, unpermute_record
    structure, x0, twin = twin_view(structure, batch.node_features)
    cfg = {"structure": structure, "batch_norms": list(model.batch_norms), "training": model.training,
           "dropout": float(model.dropout), "rng_state": getattr(model, "rng_device_state", None),
           "sync_group": sync_group_of(model), "record": model._dropout_record(),
           "grad_dst": ops.claim_destinations(params, model.training)}
    out = GcnWideEncode.apply(x0, cfg, *params)
    unpermute_record(twin, cfg.get("record"))
    return out
