"""HIP-graph capture of one whole training step (Trainer-compatible fast loop).

A step of the fused path is ~45 kernel launches; at small batches (BASELINE config 2: 512
84-ROI graphs, or one rank's 512-graph shard of the 4096-graph headline batch at 8 GPUs) the
GPU work is a few hundred microseconds but issuing it from Python costs ~1 ms.  Capturing
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
building reads sizes back to the host and stays outside the graph, like collate.  Autograd graphs
of EARLIER eager steps on the same parameters must be gone by then (do not keep their ``loss``
tensors alive; ``loss.detach()`` / ``float(loss)`` are fine): a parameter's AccumulateGrad node
lives as long as any graph that references it and stays tied to the stream it was created on, the
autograd engine then synchronises the capturing stream with that (default) stream inside the
capture, and ``capture_end`` takes the process down (segmentation fault in the HIP runtime;
torch's own warning names the cause: "The AccumulateGrad node's stream does not match ... may
break CUDA graph capture ... caused by an AccumulateGrad node created prior to the current
iteration being kept alive", tools/capture_probe.py).  ``_stale_autograd_graph`` looks for
exactly that condition with a throw-away backward on a side stream BEFORE anything is captured or
any optimisation step is taken, and the constructor raises a RuntimeError instead.

Data parallel (``grad_sync`` = dist.GradSync): gradients are views of one flat buffer that lives
outside the graph's pool, so the step is cut at its single exchange point into

    graph A: memset(flat) + forward + loss + backward      -> hipGraphLaunch
    all-reduce(flat)                                        -> one RCCL launch, same stream
    graph B: fused Adam                                     -> hipGraphLaunch

i.e. three launches per step and no capture of a collective (``collectives="split"``, the
default: it depends on nothing but stream ordering).  ``collectives="captured"`` records the
all-reduce -- and the sync-BN sum exchanges, which sit INSIDE forward/backward and therefore need
it -- into a single graph through torch's capturable NCCL/RCCL process group.
"""
from __future__ import annotations

from typing import Callable, Optional

import torch

from . import ops
from .structure import call_prepare


class GraphedTrainStep:
    """``step = GraphedTrainStep(model, optimizer, batch); loss = step()`` replays the captured
    training step on ``batch`` (a device-resident ConnectomeBatch).  ``loss`` is a static device
    tensor that every replay overwrites."""

    def __init__(self, model: torch.nn.Module, optimizer: torch.optim.Optimizer, batch,
                 loss_fn: Optional[Callable] = None, grad_sync: Optional[Callable[[], None]] = None,
                 warmup: int = 3, collectives: str = "split", local_graphs: Optional[int] = None,
                 make_batch: Optional[Callable[[], object]] = None, tolerate_capture_failure: bool = False):
        """local_graphs: with dist.GradSync and shards that may be unequal, this rank's graph count
        (the update is then the exact global-batch gradient, see dist.GradSync).

        make_batch: assemble the batch INSIDE the captured step from fixed-address inputs (``batch``
        is then only the first batch, used for the device and the warm-up): every replay re-runs the
        assembly, so one graph serves any batch of that shape -- see GraphedResidentStep.

        tolerate_capture_failure: an exception during capture (after the warm-up steps, which are real
        optimisation steps) is kept in ``capture_error`` and ``graph`` is left None instead of raising, so a
        caller that asked for "a graph where capture succeeds" can carry on eagerly from ``first_loss``."""
        if collectives not in ("split", "captured"):
            raise ValueError("collectives must be 'split' or 'captured'")
        dev = batch.node_features.device
        self.model, self.optimizer, self.batch = model, optimizer, batch
        self.make_batch = make_batch
        self.loss_fn = loss_fn or torch.nn.CrossEntropyLoss()
        self.grad_sync = grad_sync
        self._weighted = hasattr(grad_sync, "numel")       # dist.GradSync: exact with unequal shards
        self._local_graphs = None if local_graphs is None else int(local_graphs)
        if warmup < 1:
            raise ValueError("warmup >= 1: the optimizer state must exist before capture")
        split = grad_sync is not None and collectives == "split"
        if split and self._has_sync_bn():
            raise ValueError("SyncBatchNorm exchanges statistics inside forward/backward: a captured "
                             "step needs collectives='captured' (or use per-rank BatchNorm)")
        call_prepare(model.prepare_batch, batch)            # host syncs happen here, not in capture
        if self._stale_autograd_graph(dev):
            raise RuntimeError(
                "GraphedTrainStep: the autograd graph of an earlier step on these parameters is still alive "
                "(a kept, non-detached `loss` or output tensor).  Its AccumulateGrad nodes are bound to the stream "
                "that step ran on, and capturing a backward pass through them crashes the process.  Drop those "
                "tensors (`del loss`, or keep `loss.detach()` / `float(loss)`) before building a captured step.")
        if getattr(model, "rng_device_state", None) is None:
            model.rng_device_state = torch.randint(0, 2 ** 31 - 1, (16,), dtype=torch.int32, device=dev)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):                       # warm-up off the capture stream
            for _ in range(warmup):
                first = self._fwd_bwd()
                self._exchange()
                self.optimizer.step()
            # the warm-up passes are REAL optimisation steps on this batch; a caller that counts
            # them (Trainer(graph=True) uses warmup=1 as the batch's first step) reads the loss here
            self.first_loss = first.clone()
        torch.cuda.current_stream(dev).wait_stream(side)
        self._zero()
        self.graph = torch.cuda.CUDAGraph()
        self.graph_tail = None
        # with a process group alive, its watchdog thread polls events while we capture: only THIS
        # thread's calls are part of the capture
        import torch.distributed as dist
        mode = {"capture_error_mode": "thread_local"} if dist.is_available() and dist.is_initialized() else {}
        self.capture_error = None
        try:
            self._capture(split, mode)
        except Exception as exc:                            # noqa: BLE001
            if not tolerate_capture_failure:
                raise
            self.capture_error, self.graph, self.graph_tail = exc, None, None
            torch.cuda.synchronize(dev)
            for prm in model.parameters():                  # a half-captured step may have left pool-owned grads
                prm.grad = None
            if hasattr(grad_sync, "zero_grad"):
                grad_sync.zero_grad()

    def _capture(self, split: bool, mode: dict) -> None:
        # The cyclic garbage collector stays OFF while a stream is capturing: a collection pass runs in whichever
        # thread happens to allocate (here: autograd's worker thread, in the middle of the captured backward) and
        # finalises whatever cyclic garbage exists at that moment -- e.g. a dropped Trainer's captured steps, whose
        # hipGraph / private-pool teardown is not a legal call during another capture and aborts the process
        # (seen once the captured step allocated a few more Python objects and a pass landed inside the capture).
        # torch.cuda.graph() itself collects right BEFORE the capture begins; reference counting is unaffected.
        import gc
        was_enabled = gc.isenabled()
        gc.disable()
        try:
            self._capture_body(split, mode)
        finally:
            if was_enabled:
                gc.enable()

    def _capture_body(self, split: bool, mode: dict) -> None:
        if split:
            with torch.cuda.graph(self.graph, **mode):
                self.loss = self._fwd_bwd()
            self.graph_tail = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph_tail, pool=self.graph.pool(), **mode):
                self.optimizer.step()
                self._end_of_step(self.loss)
        else:
            with torch.cuda.graph(self.graph, **mode):
                self.loss = self._fwd_bwd()
                self._exchange()
                self.optimizer.step()
                self._end_of_step(self.loss)

    def capture_run(self, n: int) -> Optional["torch.cuda.CUDAGraph"]:
        """A second graph holding ``n`` CONSECUTIVE whole steps (a pool of its own), for callers that replay long
        runs of this step back to back: between two graph launches the GPU idles for the launch itself (~8 us under
        a kernel trace, 5 % of a 0.17 ms step); a run-graph pays that once per n steps.  Only where the step is one
        graph (no exchange between two graphs) and its per-step bookkeeping lives on the device (`_end_of_step`).
        None if this step cannot be captured that way (or the capture fails: the single-step graph stays in use)."""
        if self.graph is None or self.graph_tail is not None or n < 2:
            return None
        import gc
        import torch.distributed as dist
        dev = next(self.model.parameters()).device
        if self._stale_autograd_graph(dev):
            return None
        mode = {"capture_error_mode": "thread_local"} if dist.is_available() and dist.is_initialized() else {}
        g = torch.cuda.CUDAGraph()
        was_enabled = gc.isenabled()
        gc.disable()
        try:
            with torch.cuda.graph(g, **mode):
                for _ in range(n):
                    loss = self._fwd_bwd()
                    self._exchange()
                    self.optimizer.step()
                    self._end_of_step(loss)               # (each captured step's OWN loss feeds the tally)
        except Exception:                                   # noqa: BLE001 -- optional: the single-step graph remains
            torch.cuda.synchronize(dev)
            for prm in self.model.parameters():
                prm.grad = None
            if hasattr(self.grad_sync, "zero_grad"):
                self.grad_sync.zero_grad()
            return None
        finally:
            if was_enabled:
                gc.enable()
        return g

    def _stale_autograd_graph(self, dev) -> bool:
        """True if some parameter's AccumulateGrad node outlived the step that created it (an earlier
        autograd graph is being kept alive): a zero-valued backward through every parameter on a fresh side
        stream makes torch's engine report the stream mismatch (input_buffer.cpp) while nothing is being
        captured.  Gradients are restored to what they were; no parameter changes."""
        import warnings
        params = [p for p in self.model.parameters() if p.requires_grad]
        if not params:
            return False
        saved = [p.grad for p in params]
        for p in params:
            p.grad = None
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        always = torch.is_warn_always_enabled()
        torch.set_warn_always(True)                          # (the engine warns once per process otherwise)
        try:
            with warnings.catch_warnings(record=True) as caught:
                warnings.simplefilter("always")
                with torch.cuda.stream(side):
                    torch.stack([p.reshape(-1)[0] for p in params]).sum().mul(0.0).backward()
        finally:
            torch.set_warn_always(always)
            torch.cuda.current_stream(dev).wait_stream(side)
            for p, g in zip(params, saved):
                p.grad = g
        return any("AccumulateGrad node's stream does not match" in str(w.message) for w in caught)

    def _end_of_step(self, loss: torch.Tensor) -> None:
        """Captured after the optimizer step (subclasses: per-step bookkeeping on the device); ``loss`` is the
        captured step's loss tensor."""

    def _has_sync_bn(self) -> bool:
        import torch.distributed as dist
        return dist.is_initialized() and dist.get_world_size() > 1 and any(
            isinstance(m, torch.nn.SyncBatchNorm) for m in self.model.modules())

    def _zero(self) -> None:
        if hasattr(self.grad_sync, "zero_grad"):
            self.grad_sync.zero_grad()                      # one memset; .grad stay views of flat
        else:
            self.optimizer.zero_grad(set_to_none=True)

    def _fwd_bwd(self) -> torch.Tensor:
        self._zero()
        batch = self.batch if self.make_batch is None else self.make_batch()
        loss = ops.model_loss(self.model, self.loss_fn, batch)
        ops.backward_unit(loss)
        return loss.detach()

    def _exchange(self) -> None:
        if self.grad_sync is None:
            return
        if self._weighted and self._local_graphs is not None:
            self.grad_sync(local_graphs=self._local_graphs)
        else:
            self.grad_sync()

    def __call__(self) -> torch.Tensor:
        if self.graph is None:
            raise RuntimeError(f"this step was not captured: {self.capture_error!r}")
        self.graph.replay()
        if self.graph_tail is not None:
            self._exchange()
            self.graph_tail.replay()
        return self.loss


class GraphedResidentStep(GraphedTrainStep):
    """One captured step for EVERY batch of a given size drawn from a device-resident dataset with a
    per-subject structure cache (structure_cache.py: one graph per tile, the per-tile GCN path).

    The batch is assembled inside the graph -- one launch gathers node features, labels, the two
    block-offset rows and `dis` by the subject ids (cgnn_gather_rows) -- and the ids are a WINDOW of a
    fixed device buffer whose position is a device cursor that the captured step itself advances
    (cgnn_epoch_advance, which also adds loss x graphs to a device tally).  So an epoch is: copy this
    rank's permutation into the buffer once (``run_epoch``), then nothing but graph launches --
    ``Trainer(graph=True)`` over a loader that re-shuffles every epoch (the reference's semantics,
    graph.py:190-197) runs at the speed of replaying one fixed batch.  ``step(batch)`` remains for single
    batches (ids copied to the front of the buffer, cursor reset)."""

    def __init__(self, model, optimizer, first_batch, loss_fn=None, **kw):
        from .structure_cache import ResidentBatch
        from . import _lib
        cache = first_batch._cache
        dev = cache.dataset.x.device
        b = int(first_batch._ids.numel())
        self._b, self._lib = b, _lib
        self._cache_n = int(cache.n)
        self._run_graph, self._run_failed = None, False
        self.order_buf = torch.zeros(max(int(cache.dataset.num_subjects), b), dtype=torch.long, device=dev)
        self.order_buf[:b].copy_(first_batch._ids)
        # cursor (int64) and tally (fp32: sum of loss x graphs since take_tally()) share one 16-byte block, so
        # an epoch resets both with one fill
        self._state = torch.zeros(2, dtype=torch.long, device=dev)
        self.cursor = self._state[:1]
        self.tally = self._state[1:].view(torch.float32)[:1]
        self.ids_buf = self.order_buf[:b]                                   # (the window at cursor 0)
        cache.static(b)                                    # batch-size constants exist before capture
        super().__init__(model, optimizer, first_batch, loss_fn,
                         make_batch=lambda: ResidentBatch(cache, self.ids_buf, ids_offset=self.cursor), **kw)
        # the warm-up passes and the capture left cursor / tally wherever they were: start clean
        self.cursor.zero_()
        self.tally.zero_()

    def _end_of_step(self, loss: torch.Tensor) -> None:
        lib = self._lib
        dev = self.cursor.device
        with lib.device_guard(dev):
            lib.check(lib.load().cgnn_epoch_advance(lib.ptr(self.cursor), self._b, lib.ptr(loss), float(self._b),
                                                    lib.ptr(self.tally), lib.stream_ptr(dev)), "cgnn_epoch_advance")

    def __call__(self, batch=None) -> torch.Tensor:
        if batch is not None:
            # the ids as the loader handed them over: one small copy, no gather is launched
            self.ids_buf.copy_(getattr(batch, "_ids_src", batch._ids), non_blocking=True)
        self.cursor.zero_()
        return super().__call__()

    def run_epoch(self, ids: torch.Tensor, steps: int) -> None:
        """``steps`` consecutive batches of this step's size whose subject ids are ``ids`` (device,
        steps x batch_size, in order): one copy, then only replays.  Losses pile up in ``tally``."""
        n = steps * self._b
        if int(ids.numel()) != n or n > int(self.order_buf.numel()):
            raise ValueError("run_epoch: ids must hold steps x batch_size subject ids (at most the dataset's size)")
        self.order_buf[:n].copy_(ids, non_blocking=True)      # (ids may be pinned host memory: one upload per epoch)
        self._state.zero_()             # cursor and tally (single-batch calls in between also fed the tally:
                                        # their losses were read directly)
        # long runs of a SMALL step go through a graph of several consecutive steps (capture_run): the cursor and the
        # tally are advanced by the captured steps themselves, so a run-graph is just fewer launches
        k = self._run_len(steps)
        if k > 1 and self._run_graph is None and not self._run_failed:
            self._run_graph = self.capture_run(k)
            self._run_failed = self._run_graph is None
            if self._run_graph is not None:
                self._state.zero_()     # (the capture ran nothing, but keep the invariant explicit)
        done = 0
        if k > 1 and self._run_graph is not None:
            while done + k <= steps:
                self._run_graph.replay()
                done += k
        for _ in range(steps - done):
            super().__call__()

    RUN_MAX_NODES = 65536       # batches up to this many nodes (BASELINE config 2: 43,008) replay 4 steps per graph,
    RUN_MID_NODES = 262144      # up to this many (a 512 x 360 shard: 184,320) 2 steps; larger steps hide the launch

    def _run_len(self, steps: int) -> int:
        if self.graph is None or self.graph_tail is not None:
            return 1
        nodes = self._b * self._cache_n
        k = 4 if nodes <= self.RUN_MAX_NODES else (2 if nodes <= self.RUN_MID_NODES else 1)
        return k if steps >= 2 * k else 1

    def take_tally(self) -> torch.Tensor:
        """Sum of loss x graphs over the replays since the last call (device scalar); resets it."""
        out = self.tally.clone()        # (run_epoch resets the tally itself; a second take reads zero only after that)
        self.tally.zero_()
        return out.reshape(())


class GraphedEvalStep:
    """One captured EVALUATION step -- eval-mode forward (running BatchNorm statistics, no dropout), batch-mean
    loss and hit count (reference train.py:56-74) -- for every batch of a given size drawn from a
    device-resident dataset with a per-subject structure cache.  As in GraphedResidentStep the batch is
    assembled inside the graph from a window of a device id buffer at a device cursor, and the step itself
    adds loss x graphs and the hits to device tallies and advances the cursor: an evaluation pass is one copy
    of the ids, one reset, n replays and one read-back of two numbers.  Parameters and running statistics are
    read at replay time (they live at fixed addresses), so the same graph serves every epoch of training."""

    def __init__(self, model: torch.nn.Module, loss_fn: Callable, first_batch):
        from .structure_cache import ResidentBatch
        from . import _lib
        if model.training:
            raise ValueError("GraphedEvalStep captures an eval-mode forward: call model.eval() first")
        cache = first_batch._cache
        dev = cache.dataset.x.device
        b = int(first_batch._ids.numel())
        self._b, self._lib, self.model = b, _lib, model
        self.order_buf = torch.zeros(max(int(cache.dataset.num_subjects), b), dtype=torch.long, device=dev)
        self.order_buf[:b].copy_(first_batch._ids)
        self._state = torch.zeros(3, dtype=torch.long, device=dev)          # cursor | hits | loss tally (fp32)
        self.cursor, self.hits = self._state[:1], self._state[1:2]
        self.tally = self._state[2:].view(torch.float32)[:1]
        ids_buf = self.order_buf[:b]
        cache.static(b)
        make = lambda: ResidentBatch(cache, ids_buf, ids_offset=self.cursor)

        def body():
            batch = make()
            logits = model(batch)
            loss = loss_fn(logits, batch.labels)
            self.hits.add_((logits.argmax(dim=1) == batch.labels).sum())
            with _lib.device_guard(dev):
                _lib.check(_lib.load().cgnn_epoch_advance(_lib.ptr(self.cursor), b, _lib.ptr(loss), float(b),
                                                          _lib.ptr(self.tally), _lib.stream_ptr(dev)), "cgnn_epoch_advance")

        self._body, self._dev, self._nodes = body, dev, b * int(cache.n)
        self._run_graph, self._run_failed = None, False
        with torch.no_grad():
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                body()                                    # warm-up off the capture stream (builds nothing new)
            torch.cuda.current_stream(dev).wait_stream(side)
            self._state.zero_()
            self.graph = torch.cuda.CUDAGraph()
            import gc
            import torch.distributed as dist
            mode = {"capture_error_mode": "thread_local"} if dist.is_available() and dist.is_initialized() else {}
            was_enabled = gc.isenabled()
            gc.disable()                                  # (no collection pass inside a capture: GraphedTrainStep._capture)
            try:
                with torch.cuda.graph(self.graph, **mode):
                    body()
            finally:
                if was_enabled:
                    gc.enable()
        self._state.zero_()

    def run(self, ids: torch.Tensor, steps: int):
        """``steps`` consecutive batches whose subject ids are ``ids`` (steps x batch_size, device or pinned
        host): -> (sum of loss x graphs, hits) as device scalars (clones: the next run resets the tallies)."""
        n = steps * self._b
        if int(ids.numel()) != n or n > int(self.order_buf.numel()):
            raise ValueError("run: ids must hold steps x batch_size subject ids (at most the dataset's size)")
        self.order_buf[:n].copy_(ids, non_blocking=True)
        self._state.zero_()
        done = 0
        k = self.RUN_LEN if (self._nodes <= self.RUN_MAX_NODES and steps >= 2 * self.RUN_LEN) else 1
        if k > 1:
            if self._run_graph is None and not self._run_failed:
                self._capture_run(k)
            if self._run_graph is not None:
                while done + k <= steps:
                    self._run_graph.replay()
                    done += k
        for _ in range(steps - done):
            self.graph.replay()
        return self.tally.clone().reshape(()), self.hits.clone().reshape(())

    RUN_LEN, RUN_MAX_NODES = 4, 65536        # (as GraphedResidentStep: a graph of four consecutive small steps)

    def _capture_run(self, k: int) -> None:
        import gc
        import torch.distributed as dist
        mode = {"capture_error_mode": "thread_local"} if dist.is_available() and dist.is_initialized() else {}
        g = torch.cuda.CUDAGraph()
        was_enabled = gc.isenabled()
        gc.disable()
        try:
            with torch.no_grad(), torch.cuda.graph(g, **mode):
                for _ in range(k):
                    self._body()
            self._run_graph = g
        except Exception:                                   # noqa: BLE001 -- optional: single steps remain
            torch.cuda.synchronize(self._dev)
            self._run_failed = True
        finally:
            if was_enabled:
                gc.enable()
