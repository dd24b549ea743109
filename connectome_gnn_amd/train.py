"""Training loop behind the reference's ``Trainer`` API (connectome_gnn/train.py:19-127).

Constructor, ``train_epoch`` / ``evaluate`` / ``fit`` signatures, the ``history`` dict and the
metrics dict have the reference's shapes and values (golden G7 pins a three-epoch trajectory);
early stopping watches the validation loss and restores the best weights from memory.  What is
different is how the host and the GPU are kept apart:

  * nothing is read back per step: loss and hit counts pile up in device scalars
    (``_DeviceTally``) and come to the host once per epoch -- the reference's ``float(loss)``
    per step (train.py:52) would drain the HIP queue every 2.5 ms;
  * the criterion is the one-launch HIP cross-entropy (``ops.CrossEntropyLoss``), numerically
    ``torch.nn.CrossEntropyLoss()`` with default arguments;
  * the default device is "cuda" (this package has no CPU path);
  * the UNCHANGED reference script -- a list of ``ConnectomeGraph`` behind a ``ConnectomeDataLoader`` and
    ``Trainer(model, torch.optim.Adam(...), device)`` (examples/demo.py:92-134) -- does not collate on the
    host: on first sight of such a loader its dataset is packed once into HBM (``PackedDataset``), every
    epoch's permutation is drawn by the same global ``torch.randperm`` call the loader would make
    (graph.py:192-194, so seeded runs and golden G7 do not change), batches are assembled on the device
    bit-identically to ``collate_graphs``, and where the model's one-node encoder is served by the per-subject
    structure cache the step is captured once per batch size and replayed (``graph=None`` = "where capture
    succeeds"; a plain ``torch.optim.Adam`` that has not stepped yet is switched to ``capturable=True`` for
    that).  Irregular datasets (graphs of different sizes) keep the host loader;
  * data-parallel use: ``grad_sync`` (see dist.GradSync) runs between backward and the step,
    weighted by this rank's graph count so that unequal shards (partial tails) still give the
    global-batch gradient; epoch tallies (loss, hits, graphs) are summed over ranks before they
    are read, so ``history`` and the early-stopping decision are identical on every rank (a rank
    that stopped alone would leave the others hanging in their next all-reduce).
"""
from __future__ import annotations

import weakref
from collections import OrderedDict
from typing import Callable, Dict, List, Optional

import torch
import torch.nn as nn

from . import dist as cdist
from . import ops


def _joined(run) -> torch.Tensor:
    """The ids of consecutive chunks as one tensor: a view when they are adjacent slices of one array (what an
    unsharded loader yields), else a concatenation."""
    first = run[0]
    end = first.storage_offset() + first.numel()
    for c in run[1:]:
        if c.untyped_storage().data_ptr() != first.untyped_storage().data_ptr() or c.storage_offset() != end \
                or not c.is_contiguous():
            return torch.cat(run)
        end += c.numel()
    return torch.as_strided(first, (end - first.storage_offset(),), (1,))


class _DeviceTally:
    """Sum of per-batch quantities weighted by graphs per batch, kept on the device."""

    def __init__(self):
        self.total: Optional[torch.Tensor] = None
        self.graphs = 0

    def add(self, value: torch.Tensor, graphs: int = 0) -> None:
        self.total = value if self.total is None else self.total + value
        self.graphs += graphs

    def add_scaled(self, value: torch.Tensor, graphs: int) -> None:
        """total += value * graphs, one launch, `value` only read (it may be a captured step's static
        loss tensor that the next replay overwrites -- stream order keeps this read ahead of it)."""
        if self.total is None:
            self.total = value * float(graphs)
        else:
            self.total.add_(value, alpha=float(graphs))
        self.graphs += graphs

    def read(self) -> float:
        """The one device-to-host copy of an epoch."""
        return float(self.total) if self.total is not None else 0.0


class _BestWeights:
    """Early stopping on a scalar that should go down (reference train.py:104-122)."""

    def __init__(self, patience: int):
        self.patience = patience
        self.score = float("inf")
        self.epoch = 0
        self.snapshot: Optional[Dict[str, torch.Tensor]] = None

    def offer(self, epoch: int, score: float, model: nn.Module) -> None:
        if score < self.score:
            self.score, self.epoch = score, epoch
            self.snapshot = {name: t.clone() for name, t in model.state_dict().items()}

    def exhausted(self, epoch: int) -> bool:
        return epoch - self.epoch >= self.patience

    def restore(self, model: nn.Module) -> None:
        if self.snapshot is not None:
            model.load_state_dict(self.snapshot)


class Trainer:
    def __init__(self, model: nn.Module, optimizer: torch.optim.Optimizer, device: str = "cuda",
                 grad_sync: Optional[Callable[[], None]] = None, loss_fn: Optional[nn.Module] = None,
                 graph: Optional[bool] = None, graph_collectives: str = "split", max_graphs: int = 32,
                 resident: bool = True):
        """``resident`` (default on): a list-backed ``ConnectomeDataLoader`` of equally sized graphs is packed
        into HBM on first sight and iterated on the device (module docstring); ``graph=None`` (default) then
        replays a captured step where the model's encoder allows it and falls back to eager launches where
        capture fails; ``graph=False`` never captures.

        ``graph=True``: a RESIDENT batch that comes round again (same device tensors, e.g. a
        ``ResidentDataLoader(cache_batches=True)`` or any loader that yields the same device
        batches each epoch) gets its training step captured as a HIP graph
        (graphed.GraphedTrainStep) on its SECOND sighting and replayed afterwards -- one launch
        instead of ~45 for the small batches where issuing kernels from Python costs more than
        running them.  Batches of a ``ResidentDataLoader(structure_cache=True)`` are replayed from
        their FIRST sighting whatever their composition (one graph per batch size, the batch
        assembled inside it: graphed.GraphedResidentStep), so per-epoch reshuffling keeps replay
        speed.  Other batches that never repeat (a host loader whose ``.to()`` makes new tensors
        every step) run the ordinary eager step and leave nothing behind.  At
        most ``max_graphs`` steps are kept (each holds its batch and a private pool with every
        activation and gradient of the step); once that many exist, further batches stay eager.
        The optimizer must be capturable (``torch.optim.Adam(..., capturable=True)``)."""
        self.device = device
        self.model = model.to(device)
        self.optimizer = optimizer
        self.loss_fn = loss_fn if loss_fn is not None else ops.CrossEntropyLoss()
        self.grad_sync = grad_sync
        self.graph = bool(graph)
        self._graph_auto = graph is None          # decided per loader (resident + capturable), see _resident_loader
        self.resident = bool(resident)
        self._resident: Dict[int, tuple] = {}
        self.graph_collectives = graph_collectives
        self.max_graphs = int(max_graphs)
        self._graphs: Dict[tuple, object] = {}
        self._seen: "OrderedDict[tuple, weakref.ref]" = OrderedDict()
        if self.graph:
            for grp in optimizer.param_groups:
                if not grp.get("capturable", False):
                    raise ValueError("Trainer(graph=True) needs a capturable optimizer, e.g. "
                                     "torch.optim.Adam(params, ..., capturable=True)")

    # ------------------------------------------------------------- the reference script's loader, on the device
    def _resident_loader(self, loader, training: bool):
        """The device-resident twin of a list-backed ``ConnectomeDataLoader`` (same batch size, shuffle flag,
        rank / world size; its permutation comes from the same global-RNG call), built once per loader; None
        when `loader` is anything else, the device is not a GPU or the graphs are not all one size."""
        from .graph import ConnectomeDataLoader
        if not self.resident or type(loader) is not ConnectomeDataLoader:
            return None
        dev = torch.device(self.device)
        if dev.type != "cuda":
            return None
        data = loader.dataset
        if not isinstance(data, (list, tuple)) or len(data) == 0:
            return None
        sig = (id(data), len(data), id(data[0]), id(data[-1]), loader.batch_size, bool(loader.shuffle),
               loader.rank, loader.world_size)
        hit = self._resident.get(id(loader))
        if hit is not None and hit[0]() is loader and hit[1] == sig:
            rl = hit[2]
        else:
            rl = self._pack(loader, data, dev)
            self._resident[id(loader)] = (weakref.ref(loader), sig, rl)
            if len(self._resident) > 16:                       # loaders that are gone
                for key in [k for k, v in self._resident.items() if v[0]() is None]:
                    del self._resident[key]
        if rl is not None and training and self._graph_auto and not self.graph \
                and rl.structure_cache is not None and self._make_capturable():
            self.graph = True
        return rl

    def _pack(self, loader, data, dev):
        from .resident import ResidentDataLoader
        from .synthetic import PackedDataset
        g0 = data[0]
        n, e, f = g0.num_nodes, g0.num_edges, g0.num_features
        for g in data:
            lab = g.label
            if g.num_nodes != n or g.num_edges != e or g.num_features != f or not torch.is_tensor(lab) \
                    or lab.dim() != 0 or lab.dtype != torch.long or g.node_features.dtype != torch.float32 \
                    or g.edge_weight.dtype != torch.float32 or g.edge_index.dtype != torch.long:
                return None                  # irregular (or unlabelled) data: the host loader stays
        packed = PackedDataset.from_graphs(list(data))
        # torch's cross-entropy raises on a target outside [0, C) (the reference's behaviour, train.py:49); the
        # one-launch loss kernel cannot raise and turns it into a NaN loss, so the check is made here, once,
        # on the host, where the labels still are
        head = getattr(self.model, "classifier", None)
        classes = getattr(head[-1], "out_features", None) if isinstance(head, nn.Sequential) and len(head) else None
        if classes is not None and packed.labels.numel():
            bad = (packed.labels != -100) & ((packed.labels < 0) | (packed.labels >= classes))
            if bool(bad.any()):
                raise IndexError(f"Target {int(packed.labels[bad][0])} is out of bounds (labels must lie in [0, {classes}))")
        ds = packed.to(dev)
        rl = ResidentDataLoader(ds, batch_size=loader.batch_size, shuffle=bool(loader.shuffle), rank=loader.rank,
                                world_size=loader.world_size)
        if self._subject_cache_serves(ds, loader.batch_size):
            from .structure_cache import SubjectStructureCache
            rl.structure_cache = SubjectStructureCache(ds)
        return rl

    def _subject_cache_serves(self, ds, batch_size: int) -> bool:
        """The per-subject structure cache (structure_cache.py) carries what the per-tile GCN encoder and the
        GraphSAGE encoder index with, and nothing else (no CSR)."""
        from . import fused, sage_path
        from .models import GCNConnectome, GraphSAGEConnectome
        from .structure_cache import MAX_ROWS
        m = self.model
        n = int(ds.x.shape[1])
        if not (0 < n <= MAX_ROWS) or getattr(m, "impl", None) == "layered" or getattr(m, "storage", "fp32") != "fp32" \
                or any(isinstance(mod, nn.SyncBatchNorm) for mod in m.modules()) or not sage_path.bn_modules_ok(m):
            return False
        hid, fin = m.convs[0].linear.weight.shape
        if type(m) is GCNConnectome:                       # the per-tile kernels (fused.eligible)
            return hid == fused.HID and fin <= fused.MAX_F0
        if type(m) is GraphSAGEConnectome:                 # sage_path.eligible over LDS tiles
            return hid in (64, 128, 256)
        return False

    def _make_capturable(self) -> bool:
        """A captured step needs an optimizer whose step counter lives on the device.  Ours
        (optim.Adam) does; a plain torch.optim.Adam / AdamW that has not stepped yet is switched to
        ``capturable=True`` (same update, its state is created on the device at the first step)."""
        opt = self.optimizer
        groups = getattr(opt, "param_groups", [])
        if groups and all(g.get("capturable", False) for g in groups):
            return True
        if type(opt) in (torch.optim.Adam, torch.optim.AdamW) and len(opt.state) == 0 \
                and all(not g.get("differentiable", False) for g in groups):
            for g in groups:
                g["capturable"] = True
                # the multi-tensor update in ONE kernel instead of a dozen _foreach launches per step, when the
                # caller left the implementation choice to torch (both flags at their default None) and every
                # parameter is an fp32 device tensor
                if g.get("foreach") is None and g.get("fused") is None and not g.get("amsgrad", False) \
                        and all(p.is_cuda and p.dtype == torch.float32 for p in g["params"]):
                    g["fused"] = True
            return True
        return False

    def _data_parallel(self) -> bool:
        return self.grad_sync is not None and torch.distributed.is_initialized() \
            and torch.distributed.get_world_size(getattr(self.grad_sync, "group", None)) > 1

    def _global_tallies(self, loader, *tallies: "_DeviceTally"):
        """Per-epoch sums as host floats + the graph count, identical on every rank: summed over
        ranks when the loader hands each rank its own shard, rank 0's values otherwise."""
        vals = [t.total if t.total is not None else torch.zeros((), device=self.device) for t in tallies]
        if not self._data_parallel():                        # nothing to agree on: one read-back, no staging
            if len(vals) == 1:
                return [float(vals[0])], tallies[0].graphs
            return torch.stack([v.detach().double().reshape(()) for v in vals]).tolist(), tallies[0].graphs
        vec = torch.stack([v.detach().double().reshape(()) for v in vals]
                          + [torch.tensor(float(tallies[0].graphs), dtype=torch.float64, device=vals[0].device)])
        if self._data_parallel():
            group = getattr(self.grad_sync, "group", None)
            if getattr(loader, "world_size", 1) > 1:
                vec = cdist.reduce_sums(vec, group)
            else:
                torch.distributed.broadcast(vec, src=0, group=group)
        host = vec.tolist()                                    # the one read-back of the epoch
        return host[:-1], int(round(host[-1]))

    # ------------------------------------------------------------------------------ training
    @staticmethod
    def _graph_key(batch) -> tuple:
        """Identity of a resident batch.  A ResidentBatch (structure_cache.py) is keyed on its
        subject ids, never on the COO fields it assembles lazily."""
        if getattr(batch, "_ids_src", None) is not None:
            ids = batch._ids
            return ("ids", batch.node_features.data_ptr(), ids.data_ptr(), batch.labels.data_ptr(),
                    batch.num_nodes, batch.num_graphs)
        return (batch.node_features.data_ptr(), batch.edge_index.data_ptr(), batch.edge_weight.data_ptr(),
                batch.labels.data_ptr(), batch.num_nodes, batch.num_graphs, batch.edge_index._version,
                batch.edge_weight._version)

    def _graphed_step(self, batch, borrow: bool = False) -> Optional[torch.Tensor]:
        """Replay (or, on a batch's second sighting, capture) the step; None = run it eagerly."""
        if getattr(batch, "_ids_src", None) is not None and getattr(batch, "_cache", None) is not None:
            # a batch of a resident dataset with a per-subject structure cache: ONE captured step per
            # batch size serves every composition (the batch is assembled inside the graph from the
            # ids) -- fresh shuffles every epoch replay too
            rkey = ("resident", id(batch._cache), batch.num_graphs)
            step = self._graphs.get(rkey)
            if step is not None:
                return step(batch) if borrow else step(batch).clone()
            if len(self._graphs) < self.max_graphs:
                from .graphed import GraphedResidentStep
                local = batch.num_graphs if isinstance(self.grad_sync, cdist.GradSync) else None
                try:
                    step = GraphedResidentStep(self.model, self.optimizer, batch, self.loss_fn, grad_sync=self.grad_sync,
                                               warmup=1, collectives=self.graph_collectives, local_graphs=local,
                                               tolerate_capture_failure=self._graph_auto)
                except (RuntimeError, ValueError) as exc:
                    # refused BEFORE any step was taken (e.g. a live autograd graph of an earlier step)
                    if not self._graph_auto:
                        raise
                    import warnings
                    warnings.warn(f"Trainer: no step capture ({exc}); eager launches")
                    self.graph = self._graph_auto = False
                    return None
                if step.graph is None:        # graph=None ("where capture succeeds") and it did not: eager from here
                    import warnings
                    warnings.warn(f"Trainer: step capture failed ({step.capture_error!r}); eager launches")
                    self.graph = self._graph_auto = False
                    return step.first_loss    # (the warm-up pass was this batch's step)
                self._graphs[rkey] = step
                return step.first_loss        # the warm-up pass WAS this batch's step (eager)
            return None
        key = self._graph_key(batch)
        step = self._graphs.get(key)
        if step is not None:
            return step() if borrow else step().clone()
        # second sighting = the very tensor seen before is still alive (addresses alone recur: the
        # caching allocator hands a fresh batch the block its predecessor just released)
        ref = self._seen.get(key)
        if ref is None or ref() is not batch.node_features:
            self._seen[key] = weakref.ref(batch.node_features)
            self._seen.move_to_end(key)
            while len(self._seen) > 8 * max(self.max_graphs, 1):
                self._seen.popitem(last=False)
            return None
        if len(self._graphs) >= self.max_graphs:
            return None
        from .graphed import GraphedTrainStep
        local = batch.num_graphs if isinstance(self.grad_sync, cdist.GradSync) else None
        step = GraphedTrainStep(self.model, self.optimizer, batch, self.loss_fn, grad_sync=self.grad_sync,
                                warmup=1, collectives=self.graph_collectives, local_graphs=local)
        self._graphs[key] = step
        del self._seen[key]
        return step.first_loss            # the warm-up pass WAS this batch's step (eager)

    def clear_graphs(self) -> None:
        """Release every captured step (and its private memory pool)."""
        self._graphs.clear()
        self._seen.clear()

    def train_step(self, batch, _borrow: bool = False) -> torch.Tensor:
        """One optimisation step (reference train.py:46-51); returns the detached device loss.
        (_borrow: the caller consumes the loss before the next step -- a replayed step then hands
        out its static loss tensor instead of a copy.)"""
        batch = batch.to(self.device)
        if self.graph and self.model.training:
            loss = self._graphed_step(batch, _borrow)
            if loss is not None:
                return loss
        if hasattr(self.grad_sync, "zero_grad"):
            self.grad_sync.zero_grad()       # keeps .grad as views of the all-reduce buffer
        else:
            self.optimizer.zero_grad()
        loss = ops.model_loss(self.model, self.loss_fn, batch)
        ops.backward_unit(loss)
        if isinstance(self.grad_sync, cdist.GradSync):
            self.grad_sync(local_graphs=batch.num_graphs)     # exact with unequal shards
        elif self.grad_sync is not None:
            self.grad_sync()
        self.optimizer.step()
        return loss.detach()

    def _replay_epoch(self, loader, tally: "_DeviceTally") -> bool:
        """A whole epoch of a ``ResidentDataLoader(structure_cache=True)`` from captured steps: this
        rank's permutation goes to the device once, every run of equal-sized batches whose step is
        already captured is nothing but graph launches (GraphedResidentStep.run_epoch: the ids are read
        through a device cursor, the loss tally is kept by the captured step).  Batches whose size has
        no captured step yet take the ordinary path (which captures it).  False = not applicable."""
        from .resident import ResidentDataLoader
        cache = getattr(loader, "structure_cache", None)
        if not (self.graph and isinstance(loader, ResidentDataLoader) and cache is not None
                and not loader.cache_batches):
            return False
        from .structure_cache import ResidentBatch
        chunks = list(loader._chunks())
        i = 0
        while i < len(chunks):
            size = int(chunks[i].numel())
            j = i
            while j < len(chunks) and int(chunks[j].numel()) == size:
                j += 1
            step = self._graphs.get(("resident", id(cache), size))
            if step is None or self._data_parallel():
                # first sighting of this size (the ordinary path captures it); under data parallelism
                # the exchange sits between two graphs per step, which the per-batch path handles
                for c in chunks[i:j]:
                    tally.add_scaled(self.train_step(ResidentBatch(cache, c), _borrow=True), size)
            else:
                run = chunks[i:j]
                ids = run[0] if len(run) == 1 else _joined(run)
                step.run_epoch(ids.to(step.order_buf.device), len(run))
                tally.add(step.take_tally(), size * len(run))
            i = j
        return True

    def train_epoch(self, loader) -> float:
        """One pass over ``loader``; mean loss weighted by graphs per batch (train.py:52-54)."""
        self.model.train()
        tally = _DeviceTally()
        loader = self._resident_loader(loader, True) or loader
        if not self._replay_epoch(loader, tally):
            for batch in loader:
                graphs = batch.num_graphs
                tally.add_scaled(self.train_step(batch, _borrow=True), graphs)
        (loss_sum,), seen = self._global_tallies(loader, tally)
        return loss_sum / max(seen, 1)

    # ---------------------------------------------------------------------------- evaluation
    def _replay_eval(self, loader, losses: "_DeviceTally", hits: "_DeviceTally") -> bool:
        """A whole evaluation pass over a ``ResidentDataLoader(structure_cache=...)`` from captured steps
        (graphed.GraphedEvalStep, one per batch size; captured on first use).  False = not applicable: the
        caller iterates eagerly."""
        from .resident import ResidentDataLoader
        cache = getattr(loader, "structure_cache", None)
        if not (self.graph and isinstance(loader, ResidentDataLoader) and cache is not None
                and not loader.cache_batches) or self._data_parallel():
            return False
        from .graphed import GraphedEvalStep
        from .structure_cache import ResidentBatch
        chunks = list(loader._chunks())
        i = 0
        while i < len(chunks):
            size = int(chunks[i].numel())
            j = i
            while j < len(chunks) and int(chunks[j].numel()) == size:
                j += 1
            key = ("eval", id(cache), size)
            step = self._graphs.get(key)
            if step is None:
                if len(self._graphs) >= self.max_graphs:
                    step = False
                else:
                    try:
                        step = GraphedEvalStep(self.model, self.loss_fn, ResidentBatch(cache, chunks[i]))
                    except Exception as exc:              # noqa: BLE001 -- evaluation falls back to eager launches
                        if not self._graph_auto:
                            raise
                        import warnings
                        warnings.warn(f"Trainer: evaluation step not captured ({exc!r}); eager launches")
                        torch.cuda.synchronize()
                        step = False
                    self._graphs[key] = step
            run = chunks[i:j]
            if step is False:
                for c in run:
                    b = ResidentBatch(cache, c)
                    logits = self.model(b)
                    losses.add(self.loss_fn(logits, b.labels) * size, size)
                    hits.add((logits.argmax(dim=1) == b.labels).sum())
            else:
                ids = run[0] if len(run) == 1 else _joined(run)
                loss_sum, hit_sum = step.run(ids.to(step.order_buf.device), len(run))
                losses.add(loss_sum, size * len(run))
                hits.add(hit_sum)
            i = j
        return True

    @torch.no_grad()
    def evaluate(self, loader) -> dict:
        """Accuracy and mean loss (reference train.py:56-74)."""
        self.model.eval()
        losses, hits = _DeviceTally(), _DeviceTally()
        loader = self._resident_loader(loader, False) or loader
        if self._replay_eval(loader, losses, hits):
            (loss_sum, hit_sum), seen = self._global_tallies(loader, losses, hits)
            correct = int(round(hit_sum))
            return {"accuracy": correct / max(seen, 1), "loss": loss_sum / max(seen, 1),
                    "correct": correct, "total": seen}
        for batch in loader:
            batch = batch.to(self.device)
            graphs = batch.num_graphs
            logits = self.model(batch)
            losses.add(self.loss_fn(logits, batch.labels) * graphs, graphs)
            hits.add((logits.argmax(dim=1) == batch.labels).sum())
        (loss_sum, hit_sum), seen = self._global_tallies(loader, losses, hits)
        correct = int(round(hit_sum))
        return {"accuracy": correct / max(seen, 1), "loss": loss_sum / max(seen, 1),
                "correct": correct, "total": seen}

    # ----------------------------------------------------------------------------------- fit
    def fit(self, train_loader, val_loader, num_epochs: int = 50, patience: int = 10,
            verbose: bool = True) -> dict:
        """Train with early stopping on validation loss; restores the best weights
        (reference train.py:76-127).  Returns {'train_loss','val_loss','val_acc'} lists."""
        curves: Dict[str, List[float]] = {"train_loss": [], "val_loss": [], "val_acc": []}
        best = _BestWeights(patience)
        for epoch in range(1, num_epochs + 1):
            fit_loss = self.train_epoch(train_loader)
            val = self.evaluate(val_loader)
            for key, value in (("train_loss", fit_loss), ("val_loss", val["loss"]), ("val_acc", val["accuracy"])):
                curves[key].append(value)
            if verbose:
                print(f"Epoch {epoch:3d} | train_loss={fit_loss:.4f} | val_loss={val['loss']:.4f} | "
                      f"val_acc={val['accuracy']:.3f}")
            best.offer(epoch, val["loss"], self.model)
            if best.exhausted(epoch):
                if verbose:
                    print(f"Early stop at epoch {epoch} (best={best.epoch})")
                break
        best.restore(self.model)
        return curves
