"""GCN / GraphSAGE connectome classifiers on the MI355X message-passing kernels.

Drop-in for the reference's model API (connectome_gnn/models.py:159-266): same class names,
constructor signatures, attributes (``convs``, ``batch_norms``, ``classifier``, ``dropout``),
``encode``/``forward``, identical ``state_dict`` keys/shapes and -- because parameters are
created in the same order with the same initialisers -- identical initial weights under
``torch.manual_seed``.

What differs is how a layer is evaluated.  The reference issues ~13 ATen ops per GCN layer on
an int64 COO and materialises [E+N, F] message tensors (models.py:90-114).  Here a batch is
bucket-sorted once into CSR (structure.py) and a layer is
    projection   : fp32 MFMA GEMM                     (cgnn_linear_*)
    aggregation  : atomics-free segment reduction      (cgnn_aggregate_f32)
    readout      : contiguous segment mean             (cgnn_pool_mean_*)
all hand-written HIP for gfx950, reached through the C ABI in include/cgnn.h.  BatchNorm,
ReLU, dropout, the 2-layer classifier head and the loss stay PyTorch-ROCm host code in this
execution path (``impl="layered"``); the fused per-graph path replaces them as well.

There is no CPU path: CPU tensors raise (the oracle under oracle/ is test infrastructure).
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .graph import ConnectomeBatch
from .structure import BatchStructure, GcnNorm, SageNorm, _require_device


_TILE_ROWS = 384


def _grid() -> int:
    from . import _lib
    return int(_lib.load().cgnn_fused_grid())


class _AdHocBatch:
    """Lets a layer be called with raw (x, edge_index, edge_weight) like the reference's
    GCNLayer/SAGELayer (models.py:84-89,136-141): one graph, no batch vector."""

    def __init__(self, x, edge_index, edge_weight):
        n = x.shape[0]
        self.edge_index, self.edge_weight = edge_index, edge_weight
        self.batch = None
        self.ptr = torch.tensor([0, n], dtype=torch.long, device=x.device)
        self.num_nodes, self.num_graphs = n, 1


class GCNLayer(nn.Module):
    """Y = (D^-1/2 (A + I) D^-1/2)^T X W^T + b with source-side weighted degree
    (reference models.py:66-114).  Parameters: ``linear.weight [out,in]``, ``bias [out]``."""

    def __init__(self, in_channels: int, out_channels: int):
        super().__init__()
        self.linear = nn.Linear(in_channels, out_channels, bias=False)
        self.bias = nn.Parameter(torch.zeros(out_channels))
        nn.init.xavier_uniform_(self.linear.weight)

    def forward(self, x, edge_index, edge_weight, *, structure: Optional[BatchStructure] = None,
                norm: Optional[GcnNorm] = None):
        _require_device(x, "x")
        if structure is None:
            structure = BatchStructure.build(_AdHocBatch(x, edge_index, edge_weight))
        if norm is None:
            norm = structure.gcn_norm()
        s = structure
        bf, bb = s.band_ops("gcn", norm) if hasattr(s, "band_ops") else (None, None)
        fwd = (s.rowptr_dst, s.col_dst, norm.coef_dst, norm.selfc, None, bf)
        bwd = (s.rowptr_src, s.col_src, norm.coef_src, bb)
        w = self.linear.weight
        if w.shape[1] < w.shape[0]:
            # A_hat (X W^T) == (A_hat X) W^T: aggregate at the narrower width first
            return ops.linear(ops.aggregate(x, None, fwd, bwd), None, w, self.bias)
        t = ops.linear(x, None, w, None)
        if s.tiled_ok(t.shape[1]):
            # wide features: LDS-staged tiles, dis * (A_w + I)(dis * T) with the self-loop in the ELL
            meta = s.fused_meta(_TILE_ROWS, _grid(), 1.0)
            return ops.aggregate_tiled(t, self.bias, s, meta, pre=norm.dis, post=norm.dis)
        return ops.aggregate(t, self.bias, fwd, bwd)


class SAGELayer(nn.Module):
    """relu(Linear([x || weighted-mean of in-neighbours])) (reference models.py:121-152)."""

    def __init__(self, in_channels: int, out_channels: int):
        super().__init__()
        self.linear = nn.Linear(in_channels * 2, out_channels)
        nn.init.xavier_uniform_(self.linear.weight)

    def forward(self, x, edge_index, edge_weight, *, structure: Optional[BatchStructure] = None,
                norm: Optional[SageNorm] = None):
        _require_device(x, "x")
        if structure is None:
            structure = BatchStructure.build(_AdHocBatch(x, edge_index, edge_weight))
        if norm is None:
            norm = structure.sage_norm()
        s = structure
        if s.tiled_ok(x.shape[1]):
            # wide features: LDS-staged tiles, (A_w X) / (wsum + 1e-8), no self-loop
            meta = s.fused_meta(_TILE_ROWS, _grid(), 0.0)
            agg = ops.aggregate_tiled(x, None, s, meta, post=norm.den, post_div=True)
        else:
            bf, bb = s.band_ops("sage", norm) if hasattr(s, "band_ops") else (None, None)
            agg = ops.aggregate(x, None, (s.rowptr_dst, s.col_dst, norm.w_dst, None, norm.den, bf),
                                (s.rowptr_src, s.col_src, norm.coef_src_bwd, bb))
        # the [x || agg] concat is never materialised: two K-panels of one GEMM, ReLU epilogue
        return ops.linear(x, agg, self.linear.weight, self.linear.bias, relu=True)


def _head(hidden_dim: int, num_classes: int, dropout: float) -> nn.Sequential:
    return nn.Sequential(nn.Linear(hidden_dim, hidden_dim // 2), nn.ReLU(), nn.Dropout(dropout),
                         nn.Linear(hidden_dim // 2, num_classes))


class _ConnectomeModel(nn.Module):
    _layer_cls = None
    _relu_after_bn = False

    def __init__(self, in_channels: int, hidden_dim: int = 64, num_classes: int = 2,
                 num_layers: int = 3, dropout: float = 0.3, *, impl: str = "auto", storage: str = "fp32"):
        super().__init__()
        if impl not in ("auto", "fused", "layered"):
            raise ValueError("impl must be 'auto', 'fused' or 'layered'")
        if storage not in ("fp32", "fp16"):
            raise ValueError("storage must be 'fp32' (the reference's arithmetic) or 'fp16'")
        self.dropout = dropout
        self.impl = impl              # execution path; not part of the reference API/state_dict
        # "fp16": activations cross HBM as IEEE half, fp32 accumulate (GCN, large dense graphs:
        # gcn_half_path.py); parameters, gradients and the state_dict stay fp32
        self.storage = storage
        if storage == "fp16" and not self._relu_after_bn:
            raise ValueError("storage='fp16' is implemented for GCNConnectome only")
        if storage == "fp16" and in_channels < 2:
            # measured: with ONE input feature layer 0 is rank one, every channel of its
            # BatchNorm output is the same signal up to sign, and half-rounded activations put up to 10 % on the
            # classifier's weight gradient (fp32 storage: 1e-6 on the same inputs)
            raise ValueError("storage='fp16' needs at least 2 input features (a rank-one layer 0 loses the "
                             "classifier gradient to half rounding); use storage='fp32'")
        self.impl_used = None
        self.rng_device_state = None  # uint32 device words for graph-captured dropout (graphed.py)
        # parity hook: when True, every training forward leaves the dropout keep decisions it drew
        # in ``last_dropout`` ({"layers": [uint8 keep bits per layer], "head_factor": [B,H/2]}), so a
        # test can replay the same masks through the CPU oracle (tests/test_gpu_models.py)
        self.record_dropout = False
        self.last_dropout = None
        widths = [in_channels] + [hidden_dim] * num_layers
        self.convs = nn.ModuleList(self._layer_cls(a, b) for a, b in zip(widths, widths[1:]))
        self.batch_norms = nn.ModuleList(nn.BatchNorm1d(hidden_dim) for _ in range(num_layers))
        self.classifier = _head(hidden_dim, num_classes, dropout)

    def _norm(self, structure: BatchStructure):
        raise NotImplementedError

    def _post(self, x):
        raise NotImplementedError

    def _dropout_record(self):
        """Fresh record dict for this forward when ``record_dropout`` is on, else None."""
        if not (self.record_dropout and self.training and self.dropout > 0):
            return None
        if self.last_dropout is None or "layers" in self.last_dropout:
            self.last_dropout = {}
        return self.last_dropout

    def _validate(self, batch: ConnectomeBatch) -> None:
        """The fused encoders hand raw device pointers to the kernels, so what the reference's
        F.linear would reject (models.py:111,151: fp16/fp64 tensors, a feature width that is not
        in_channels) is rejected here, before any path is chosen."""
        x = batch.node_features
        _require_device(x, "batch.node_features")
        if x.dtype != torch.float32:
            raise TypeError(f"batch.node_features must be float32 (the reference is fp32-only), got {x.dtype}")
        for name, prm in self.named_parameters():
            if prm.dtype != torch.float32:
                raise TypeError(f"parameter {name} must be float32, got {prm.dtype}")
            if not prm.is_cuda:
                _require_device(prm, f"parameter {name}")
        fin = self.convs[0].linear.weight.shape[1] // (1 if self._relu_after_bn else 2)
        if x.dim() != 2 or x.shape[1] != fin:
            raise ValueError(f"batch.node_features must be [num_nodes, {fin}] (in_channels), got "
                             f"{tuple(x.shape)}")

    def encode(self, batch: ConnectomeBatch) -> torch.Tensor:
        """Graph embeddings [B, hidden] (reference models.py:203-211 / 256-262)."""
        self._validate(batch)
        s = batch.structure()
        if self._agreed_fused(batch, s):
            self.impl_used = "fused"
            return self._fused_encode(batch, s)
        self.impl_used = "layered"
        norm = self._norm(s)                  # once per forward pass, shared by all layers
        x = batch.node_features
        rng = getattr(self, "rng_device_state", None)     # set by graphed.GraphedTrainStep
        if rng is not None and self.training and self.dropout > 0:
            from . import _lib
            _lib.check(_lib.load().cgnn_rng_advance(_lib.ptr(rng), len(self.convs) + 1,
                                                    _lib.stream_ptr()), "cgnn_rng_advance")
        rec = self._dropout_record()
        for li, (conv, bn) in enumerate(zip(self.convs, self.batch_norms)):
            x = conv(x, batch.edge_index, batch.edge_weight, structure=s, norm=norm)
            if ops.bn_act_drop_supported(bn, x.shape[1]):
                # BatchNorm (+ReLU) + dropout in two streaming HIP passes each way
                x = ops.bn_act_drop(x, bn, self._relu_after_bn, self.dropout, self.training,
                                    None if rng is None else rng.data_ptr() + 4 * li, rec)
            else:                     # SyncBatchNorm / odd widths: torch ops
                x = self._post(bn(x))
                x = F.dropout(x, p=self.dropout, training=self.training)
        return ops.pool_mean(x, s.gptr, batch.num_graphs)

    def forward_loss(self, batch: ConnectomeBatch):
        """(logits, mean cross-entropy against batch.labels) with the classifier, the loss and their
        backward in ONE launch (ops.head_loss) -- the Trainer's step, reference train.py:48-50 -- or None
        when the head is not one of that kernel's shapes or the model is not training on a GPU batch."""
        if not (self.training and batch.node_features.is_cuda and batch.labels is not None
                and ops.head_loss_supported(self.classifier)):
            return None
        pooled = self.encode(batch)
        rng = getattr(self, "rng_device_state", None)
        word = None if rng is None else rng.data_ptr() + 4 * len(self.convs)
        rec = self.last_dropout if self.record_dropout else None
        return ops.head_loss(self.classifier, pooled, batch.labels, True, word, rec)

    def forward(self, batch: ConnectomeBatch) -> torch.Tensor:
        """Class logits [B, num_classes]."""
        pooled = self.encode(batch)
        if ops.head_supported(self.classifier):
            # Linear -> ReLU -> Dropout -> Linear on [B, hidden] in one HIP kernel each way
            rng = getattr(self, "rng_device_state", None)
            word = None if rng is None else rng.data_ptr() + 4 * len(self.convs)
            rec = self.last_dropout if (self.record_dropout and self.training) else None
            return ops.head(self.classifier, pooled, self.training, word, rec)
        if self.record_dropout and self.training and self.last_dropout is not None \
                and self.dropout > 0 and len(self.classifier) == 4:
            # parity hook for heads outside head.hip's widths (torch modules): same record format
            l1, act, drop, l2 = self.classifier
            h = act(l1(pooled))
            y = drop(h)
            self.last_dropout["head_factor"] = ((y != 0) & (h > 0)).float() / (1.0 - drop.p)
            return l2(y)
        return self.classifier(pooled)

    def prepare_batch(self, batch: ConnectomeBatch, reuse: bool = False) -> None:
        """Build every piece of static per-batch metadata this model will use (CSR, and the
        blocked-ELL of the fused path) now -- these builds read sizes back to the host, so they
        must not happen inside a HIP-graph capture or a timed region.  reuse=True says the batch
        will be trained on repeatedly (cached batches, captured steps): structure work that only
        pays when amortised is then done too (the per-tile GCN path's degree-ordered twin)."""
        s = batch.structure()
        if reuse:
            self._prepare_reused(batch, s)
            if getattr(s, "__dict__", {}).get("_degree_twin") is not None:
                return                                  # the encoder runs on the twin's metadata
        if self.storage == "fp16":
            from . import gcn_half_path
            gcn_half_path.dense_operators(s)
            return
        hid = self.convs[-1].linear.weight.shape[0]
        if s.tiled_ok(hid):
            # GCN's ELL carries the self-loop (weight 1), GraphSAGE's does not (weight 0)
            s.fused_meta(_TILE_ROWS, _grid(), 1.0 if self._relu_after_bn else 0.0)
        elif hasattr(s, "band_ops"):
            # large graphs: the dense fragments of the operator as matrix-core operands (band_aggregate.hip)
            s.band_ops("gcn" if self._relu_after_bn else "sage", self._norm(s))
        self._try_fused(batch, s)

    def _try_fused(self, batch, structure) -> bool:
        return False

    def _prepare_reused(self, batch, structure) -> None:
        """prepare_batch(reuse=True): the one-node encoders over LDS tiles (per-tile GCN, wide GCN,
        GraphSAGE) run on the batch's degree-ordered twin (structure.degree_ordered_twin: 19 % -> 3 % of
        blocked-ELL padding on small-world connectomes).  Not under cross-rank BatchNorm (the ranks
        agree on paths there), not for cached subject structures, not for the dense fp16 operator."""
        from .structure import BatchStructure
        if not isinstance(structure, BatchStructure) or self.impl == "layered" or self.storage == "fp16" \
                or any(isinstance(bn, nn.SyncBatchNorm) for bn in self.batch_norms) \
                or not self._try_fused(batch, structure) \
                or not structure.tiled_ok(self.convs[-1].linear.weight.shape[0]):
            return
        twin = structure.degree_ordered_twin()
        twin.fused_meta(_TILE_ROWS, _grid(), 1.0 if self._relu_after_bn else 0.0)
        twin.permuted_features(batch.node_features)

    def _agreed_fused(self, batch, structure) -> bool:
        """``_try_fused`` -- and, while training with SyncBatchNorm across ranks, the same answer
        on every rank: the choice depends on the local shard (graph sizes, block-diagonality),
        and the fused encoders exchange their BatchNorm sums with a different collective than
        torch's SyncBatchNorm on the layered path, so mixed paths would hang.  One tiny
        all-reduce(MIN) per batch, cached on the batch's structure."""
        ok = self._try_fused(batch, structure)
        import torch.distributed as dist
        if not (self.training and dist.is_initialized() and dist.get_world_size() > 1
                and any(isinstance(bn, nn.SyncBatchNorm) for bn in self.batch_norms)):
            return ok
        key = (type(self).__name__, self.impl, id(self))
        cache = structure.__dict__.setdefault("_path_agreement", {})
        if key not in cache:
            from . import dist as cdist
            code = 0 if not ok else (2 if getattr(self, "_fused_kind", "tile") == "tile" else 1)
            cache[key] = cdist.agree_min(code, batch.node_features.device)
        agreed = cache[key]
        if ok and agreed == 0:
            return False
        if ok and agreed == 1:
            self._fused_kind = "wide"
        return ok

    def _fused_encode(self, batch, structure) -> torch.Tensor:
        raise NotImplementedError

    def _decide(self, why: Optional[str]) -> bool:
        if why is not None and self.impl == "fused":
            raise RuntimeError(f"impl='fused' requested but not applicable: {why}")
        return why is None


class GCNConnectome(_ConnectomeModel):
    """conv -> BatchNorm1d -> ReLU -> dropout per layer, mean-pool, MLP head
    (reference models.py:159-216)."""
    _layer_cls = GCNLayer
    _relu_after_bn = True

    def _norm(self, structure):
        return structure.gcn_norm()

    def _post(self, x):
        return F.relu(x)

    def _try_fused(self, batch, structure) -> bool:
        """Fused per-tile kernels (fused.py) when the shape is covered: hidden 64, <= 16 input
        features, graphs of <= 384 nodes, block-diagonal edges; else the one-node wide encoder
        (gcn_wide_path.py) for hidden 64/128/256 on such graphs; else the op-by-op path."""
        if self.storage == "fp16":
            from . import gcn_half_path
            why = gcn_half_path.eligible(self, batch, structure)
            if why is not None:
                raise RuntimeError(f"storage='fp16' requested but not applicable: {why}")
            self._fused_kind = "half"
            return True
        if self.impl == "layered":
            return False
        from . import fused, gcn_wide_path
        why = fused.eligible(self, batch, structure)
        self._fused_kind = "tile"
        if why is not None and gcn_wide_path.eligible(self, batch, structure) is None:
            self._fused_kind, why = "wide", None
        return self._decide(why)

    def _fused_encode(self, batch, structure):
        from . import fused, gcn_half_path, gcn_wide_path
        path = {"tile": fused, "wide": gcn_wide_path, "half": gcn_half_path}[self._fused_kind]
        return path.encode(self, batch, structure)

class GraphSAGEConnectome(_ConnectomeModel):
    """conv (ReLU inside) -> BatchNorm1d -> dropout per layer -- no ReLU after BN
    (reference models.py:219-266)."""
    _layer_cls = SAGELayer

    def _norm(self, structure):
        return structure.sage_norm()

    def _post(self, x):
        return x

    def _try_fused(self, batch, structure) -> bool:
        """One-node encoder with hand-written backward (sage_path.py) when hidden is a multiple
        of 64 and every graph fits an LDS tile (<= 384 nodes, block-diagonal edges)."""
        if self.impl == "layered":
            return False
        from . import sage_path
        return self._decide(sage_path.eligible(self, batch, structure))

    def _fused_encode(self, batch, structure):
        from . import sage_path
        return sage_path.encode(self, batch, structure)
