"""HIP-graph capture of one whole training step (Trainer-compatible fast loop).

A step of the fused path is ~45 kernel launches; at small batches (BASELINE config 2: 512
84-ROI graphs) the GPU work is ~0.15 ms but issuing it from Python costs ~1 ms.  Capturing
zero_grad + forward + loss + backward + optimizer step once per resident batch and replaying
it turns the host cost into one hipGraphLaunch.

What makes the step capturable:
  * the HIP library never allocates or synchronises (include/cgnn.h conventions); its kernels are
    launched on torch's current stream, which is the capturing stream;
  * torch.empty inside capture comes from the graph's private pool;
  * dropout seeds are by-value kernel arguments, i.e. frozen in the graph -- so every mask kernel
    also XORs a device word that a captured ``cgnn_rng_advance`` refreshes on each replay
    (``model.rng_device_state``);
  * BatchNorm's ``num_batches_tracked`` is bumped on the device by the finalise kernel.
The batch must be resident with its structure (CSR / blocked-ELL) already built: structure
building reads sizes back to the host and stays outside the graph, like collate.
"""
from __future__ import annotations

from typing import Callable, Optional

import torch


class GraphedTrainStep:
    """``step = GraphedTrainStep(model, optimizer, batch); loss = step()`` replays the captured
    training step on ``batch`` (a device-resident ConnectomeBatch).  ``loss`` is a static device
    tensor that every replay overwrites."""

    def __init__(self, model: torch.nn.Module, optimizer: torch.optim.Optimizer, batch,
                 loss_fn: Optional[Callable] = None, grad_sync: Optional[Callable[[], None]] = None,
                 warmup: int = 3):
        if grad_sync is not None:
            raise NotImplementedError("graph capture with a gradient all-reduce is not wired yet; "
                                      "use the eager step for multi-rank training")
        dev = batch.node_features.device
        self.model, self.optimizer, self.batch = model, optimizer, batch
        self.loss_fn = loss_fn or torch.nn.CrossEntropyLoss()
        if warmup < 1:
            raise ValueError("warmup >= 1: the optimizer state must exist before capture")
        model.prepare_batch(batch)                          # host syncs happen here, not in capture
        if getattr(model, "rng_device_state", None) is None:
            model.rng_device_state = torch.randint(0, 2 ** 31 - 1, (16,), dtype=torch.int32, device=dev)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):                       # warm-up off the capture stream
            for _ in range(warmup):
                self._eager()
        torch.cuda.current_stream(dev).wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        optimizer.zero_grad(set_to_none=True)
        with torch.cuda.graph(self.graph):
            self.loss = self._eager()

    def _eager(self) -> torch.Tensor:
        self.optimizer.zero_grad(set_to_none=True)
        loss = self.loss_fn(self.model(self.batch), self.batch.labels)
        loss.backward()
        self.optimizer.step()
        return loss.detach()

    def __call__(self) -> torch.Tensor:
        self.graph.replay()
        return self.loss
