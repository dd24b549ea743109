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
  * data-parallel use: ``grad_sync`` (see dist.GradSync) runs between backward and the step.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional

import torch
import torch.nn as nn

from . import ops


class _DeviceTally:
    """Sum of per-batch quantities weighted by graphs per batch, kept on the device."""

    def __init__(self):
        self.total: Optional[torch.Tensor] = None
        self.graphs = 0

    def add(self, value: torch.Tensor, graphs: int = 0) -> None:
        self.total = value if self.total is None else self.total + value
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
                 grad_sync: Optional[Callable[[], None]] = None):
        self.device = device
        self.model = model.to(device)
        self.optimizer = optimizer
        self.loss_fn = ops.CrossEntropyLoss()
        self.grad_sync = grad_sync

    # ------------------------------------------------------------------------------ training
    def train_step(self, batch) -> torch.Tensor:
        """One optimisation step (reference train.py:46-51); returns the detached device loss."""
        batch = batch.to(self.device)
        if hasattr(self.grad_sync, "zero_grad"):
            self.grad_sync.zero_grad()       # keeps .grad as views of the all-reduce buffer
        else:
            self.optimizer.zero_grad()
        loss = self.loss_fn(self.model(batch), batch.labels)
        loss.backward()
        if self.grad_sync is not None:
            self.grad_sync()
        self.optimizer.step()
        return loss.detach()

    def train_epoch(self, loader) -> float:
        """One pass over ``loader``; mean loss weighted by graphs per batch (train.py:52-54)."""
        self.model.train()
        tally = _DeviceTally()
        for batch in loader:
            graphs = batch.num_graphs
            tally.add(self.train_step(batch) * graphs, graphs)
        return tally.read() / max(tally.graphs, 1)

    # ---------------------------------------------------------------------------- evaluation
    @torch.no_grad()
    def evaluate(self, loader) -> dict:
        """Accuracy and mean loss (reference train.py:56-74)."""
        self.model.eval()
        losses, hits = _DeviceTally(), _DeviceTally()
        for batch in loader:
            batch = batch.to(self.device)
            graphs = batch.num_graphs
            logits = self.model(batch)
            losses.add(self.loss_fn(logits, batch.labels) * graphs, graphs)
            hits.add((logits.argmax(dim=1) == batch.labels).sum())
        seen = losses.graphs
        correct = int(hits.read())
        return {"accuracy": correct / max(seen, 1), "loss": losses.read() / max(seen, 1),
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
