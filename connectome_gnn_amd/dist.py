"""Graph-sharded data parallelism: one process per GPU, RCCL over xGMI.

Graphs are independent through conv/pool/classifier, so a global batch is cut into contiguous
runs of graphs, one per rank; nothing on the data path is exchanged.  What IS exchanged:

  * gradients -- one flat fp32 buffer (11,234 floats = 45 KB for GCN h=64), averaged with a
    single all-reduce per step.  CrossEntropy is a mean over graphs, so with equal shards
    mean-of-means is the global-batch gradient; with unequal shards each rank weights its
    gradient by its share of the graphs first.
  * (optional) BatchNorm statistics -- the reference normalises over ALL nodes of the global
    batch; ``convert_sync_batchnorm`` keeps that semantics across ranks.

The collectives are latency-bound (tens of KB over 7 x 153 GB/s links), so there is exactly one
gradient all-reduce per step, issued on the compute stream right after backward.
``backend="nccl"`` is RCCL on ROCm; tests run the same code over gloo on CPU tensors.
"""
from __future__ import annotations

import os
from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None) -> tuple:
    """Join the process group torchrun set up (RANK/WORLD_SIZE/LOCAL_RANK/MASTER_*).
    Returns (rank, world_size, local_rank).  Single process -> (0, 1, 0) with no group."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


class GradSync:
    """Average gradients over ranks with ONE all-reduce of a persistent flat buffer.

    Every parameter's ``.grad`` is a permanent VIEW into the flat buffer (like DDP's
    gradient_as_bucket_view), so a step costs one memset (``zero_grad``), backward accumulating
    in place, and one all-reduce -- no flatten/unflatten copy kernels.  Use
    ``sync.zero_grad()`` instead of ``optimizer.zero_grad()`` (which would drop the views).

    Unequal shards (a global batch that does not divide by the world size -- every partial tail
    of the sharded loaders): ``sync(local_graphs=n_r)`` turns the update into the exact
    global-batch gradient  sum_r n_r g_r / sum_r n_r.  The count rides in one extra word at the
    end of the flat buffer, so it is still a single collective and nothing returns to the host
    (the whole step stays capturable in a HIP graph).
    """

    def __init__(self, params: Iterable[torch.nn.Parameter], group=None, direct: bool = True):
        """direct: after every ``zero_grad()`` the package's one-node encoders and its classifier + loss
        launch write their parameter gradients STRAIGHT into the (just zeroed) views instead of handing
        them to autograd, whose AccumulateGrad would add each one onto its view with a kernel of its own
        -- 16 launches per step for the 3-layer GCN, 35 us of a 0.31 ms step at 512 graphs per rank
        (ops.grad_destination; assumes every parameter feeds ONE op, which holds for the reference's
        models).  Gradients produced by ordinary torch ops still accumulate through autograd."""
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        self.group = group
        self.direct = bool(direct)
        n = sum(p.numel() for p in self.params)
        ref = self.params[0]
        self.numel = n
        self._buf = torch.zeros(n + 1, dtype=ref.dtype, device=ref.device)   # [grads | graph count]
        self.flat = self._buf[:n]
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self._avg = dist.is_initialized() and dist.get_backend(group) == "nccl"   # RCCL has AVG
        self._attach()

    def _attach(self) -> None:
        off = 0
        for p in self.params:
            n = p.numel()
            p.grad = self.flat[off:off + n].view_as(p)
            off += n

    def zero_grad(self) -> None:
        """One memset; re-attaches a view if something replaced a ``.grad`` (e.g. set_to_none)."""
        self.flat.zero_()
        off = 0
        for p in self.params:
            n = p.numel()
            g = p.grad
            if g is None or g.data_ptr() != self.flat.data_ptr() + off * self.flat.element_size():
                p.grad = self.flat[off:off + n].view_as(p)
            if self.direct:
                p._cgnn_direct = True          # armed: the view is zero, one op may write its gradient into it
            off += n

    def __call__(self, local_graphs: Optional[int] = None, global_graphs: Optional[int] = None):
        """Call between ``loss.backward()`` and ``optimizer.step()``.  ``local_graphs``: this
        rank's graph count when shards may be unequal (``global_graphs`` is accepted for
        compatibility and ignored: the total is summed by the same all-reduce)."""
        if self.world == 1:
            return
        off = 0
        for p in self.params:          # a grad that autograd replaced (not accumulated in place)
            n = p.numel()
            g = p.grad
            if g is not None and g.data_ptr() != self.flat.data_ptr() + off * self.flat.element_size():
                self.flat[off:off + n].copy_(g.reshape(-1))
                p.grad = self.flat[off:off + n].view_as(p)
            off += n
        if local_graphs is not None:
            self.flat.mul_(float(local_graphs))
            self._buf[self.numel:].fill_(float(local_graphs))
            dist.all_reduce(self._buf, op=dist.ReduceOp.SUM, group=self.group)
            self.flat.div_(self._buf[self.numel:])
        elif self._avg:
            try:
                dist.all_reduce(self.flat, op=dist.ReduceOp.AVG, group=self.group)
            except (RuntimeError, ValueError):
                # a backend build without the AVG reduction: scale + SUM from now on (the failed call
                # raised before it enqueued anything, so the buffer still holds this rank's gradient)
                self._avg = False
                self.flat.mul_(1.0 / self.world)
                dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
        else:
            self.flat.mul_(1.0 / self.world)
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)


def agree_min(code: int, device, group=None) -> int:
    """The smallest ``code`` over all ranks (one tiny all-reduce + read-back; callers cache the
    answer per batch).  Used to keep all ranks on the same execution path: under sync-BN the
    fused and the layered encoders issue different collectives, and a mismatch would hang."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return int(code)
    dev = device if dist.get_backend(group) == "nccl" else "cpu"
    t = torch.tensor([int(code)], dtype=torch.int32, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return int(t.item())


def agree(flag: bool, device, group=None) -> bool:
    """True iff ``flag`` is True on EVERY rank."""
    return bool(agree_min(1 if flag else 0, device, group))


def reduce_sums(values: torch.Tensor, group=None) -> torch.Tensor:
    """Sum a small tally vector over ranks (loss / hit / graph counts of an epoch)."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(values, op=dist.ReduceOp.SUM, group=group)
    return values


def broadcast_parameters(module: torch.nn.Module, src: int = 0, group=None) -> None:
    """Make every rank start from rank ``src``'s parameters and buffers."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src, group=group)


def convert_sync_batchnorm(module: torch.nn.Module, group=None) -> torch.nn.Module:
    """Full-batch BatchNorm statistics across ranks (exact parity with the single-process
    oracle at N_gpu > 1).  Uses torch's SyncBatchNorm: 2H+1 floats forward, 2H backward per
    layer over RCCL."""
    return torch.nn.SyncBatchNorm.convert_sync_batchnorm(module, process_group=group)
