"""Training loop with the reference's ``Trainer`` API (connectome_gnn/train.py:19-127).

Same constructor, ``train_epoch``/``evaluate``/``fit`` signatures, ``history`` and metrics
dict shapes, early stopping on validation loss with in-memory best-state restore.  Two
deliberate differences, both host-side:
  * the default device is "cuda" (there is no CPU path in this package);
  * losses/correct counts are accumulated on the device and read back once per epoch, not
    once per step (the reference's ``float(loss)`` at train.py:52 stalls the HIP queue).
Data-parallel use: pass ``grad_sync`` (see dist.GradSync) -- called between backward and step.
"""
from __future__ import annotations

from typing import Callable, Optional

import torch
import torch.nn as nn

from . import ops


class Trainer:
    def __init__(self, model: nn.Module, optimizer: torch.optim.Optimizer, device: str = "cuda",
                 grad_sync: Optional[Callable[[], None]] = None):
        self.model = model.to(device)
        self.optimizer = optimizer
        self.device = device
        self.loss_fn = ops.CrossEntropyLoss()       # == nn.CrossEntropyLoss(), one HIP launch
        self.grad_sync = grad_sync

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
        weighted, seen = None, 0
        for batch in loader:
            loss = self.train_step(batch) * batch.num_graphs
            weighted = loss if weighted is None else weighted + loss
            seen += batch.num_graphs
        return float(weighted) / max(seen, 1) if weighted is not None else 0.0

    @torch.no_grad()
    def evaluate(self, loader) -> dict:
        """Accuracy and mean loss (reference train.py:56-74)."""
        self.model.eval()
        weighted = hits = None
        seen = 0
        for batch in loader:
            batch = batch.to(self.device)
            logits = self.model(batch)
            loss = self.loss_fn(logits, batch.labels) * batch.num_graphs
            ok = (logits.argmax(dim=1) == batch.labels).sum()
            weighted = loss if weighted is None else weighted + loss
            hits = ok if hits is None else hits + ok
            seen += batch.num_graphs
        correct = int(hits) if hits is not None else 0
        total_loss = float(weighted) if weighted is not None else 0.0
        return {"accuracy": correct / max(seen, 1), "loss": total_loss / max(seen, 1),
                "correct": correct, "total": seen}

    def fit(self, train_loader, val_loader, num_epochs: int = 50, patience: int = 10,
            verbose: bool = True) -> dict:
        """Train with early stopping on validation loss; restores the best weights
        (reference train.py:76-127).  Returns {'train_loss','val_loss','val_acc'} lists."""
        history = {"train_loss": [], "val_loss": [], "val_acc": []}
        best_loss, best_epoch, best_state = float("inf"), 0, None
        for epoch in range(1, num_epochs + 1):
            tl = self.train_epoch(train_loader)
            ev = self.evaluate(val_loader)
            history["train_loss"].append(tl)
            history["val_loss"].append(ev["loss"])
            history["val_acc"].append(ev["accuracy"])
            if verbose:
                print(f"Epoch {epoch:3d} | train_loss={tl:.4f} | val_loss={ev['loss']:.4f} | "
                      f"val_acc={ev['accuracy']:.3f}")
            if ev["loss"] < best_loss:
                best_loss, best_epoch = ev["loss"], epoch
                best_state = {k: v.clone() for k, v in self.model.state_dict().items()}
            if epoch - best_epoch >= patience:
                if verbose:
                    print(f"Early stop at epoch {epoch} (best={best_epoch})")
                break
        if best_state is not None:
            self.model.load_state_dict(best_state)
        return history
