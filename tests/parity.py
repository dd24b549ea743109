"""Shared parity machinery of the -m gpu tests: run the oracle in fp32 and fp64 on the same
inputs, compare a HIP result against it, and keep score of how often the fp64 arbiter decided.

Gradient tolerance: a gradient tensor passes if
  (1) it is within 1e-5 of the tensor's scale of the fp32 oracle (north_star's 1e-5:
      max|got - want| <= 1e-5 * max|want| + 2e-6), or elementwise within rtol 1e-4 / atol 2e-6
      -- the bars the oracle itself is held to against the goldens recorded from the reference
      (tests/test_oracle_golden.py); or
  (2) the fp64 ARBITER: its max error against the oracle evaluated in float64 is no larger than
      the fp32 oracle's own max error against float64 (factor 1, plus 1e-9 absolute).  Rule (2)
      exists because the reference's fp32 CPU arithmetic is sometimes the noisy side: a ReLU
      pre-activation within rounding of 0, or BatchNorm over a near-constant column (GCN over
      dense graphs smooths node features until a channel's variance is rounding-sized), moves a
      weight gradient by 1e-5..1e-3 of its scale in ANY fp32 evaluation order.
  (3) the fp32 NOISE FLOOR, tried last: "no worse than ONE fp32 evaluation" is a coin flip when
      both sides are equally noisy, so the reference's own fp32 spread is sampled: the oracle is
      re-run in fp32 on the same batch presented in other orders (graphs reversed, COO edge
      order reversed, both -- a batch is a set of graphs and a COO a set of edges, so parameter
      gradients are mathematically unchanged and each run is as much "the reference's answer"
      as the first).  The HIP gradient passes if its error against fp64 is within the largest
      error of those evaluations (factor 1).
Every decision is tallied in ARBITER; tests/test_zz_arbiter.py prints the tally and fails if rules
(2)+(3) decided more than 5 % of the tensors, so they cannot silently become the norm.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from oracle import reference_path as O

TOL = dict(rtol=1e-5, atol=2e-6)
GTOL = dict(rtol=1e-4, atol=2e-6)
ARBITER = {"checked": 0, "fp64": 0, "floor": 0, "names": []}


def oracle_run(kind: str, state_dict: Dict[str, torch.Tensor], b, dropout: float = 0.0,
               training: bool = True, masks: Optional[dict] = None, dtype=torch.float32):
    """One forward + CE + backward of the oracle on CPU copies, in ``dtype``.
    Returns (logits, loss, {param: grad}, state after the step)."""
    torch.set_default_dtype(dtype)
    try:
        cast = lambda v: v.detach().cpu().clone().to(dtype) if v.is_floating_point() else v.detach().cpu().clone()
        st = O.require_grad({k: cast(v) for k, v in state_dict.items()})
        ob = O.OBatch(cast(b.node_features), b.edge_index.cpu(), cast(b.edge_weight), b.batch.cpu(),
                      b.labels.cpu(), b.ptr.cpu())
        mk = None
        if masks is not None:
            mk = {"layers": [cast(m) for m in masks["layers"]], "head": cast(masks["head"])}
        logits = O.FORWARD[kind](st, ob, dropout, training, mk)
        loss = torch.nn.functional.cross_entropy(logits, ob.labels)
        loss.backward()
    finally:
        torch.set_default_dtype(torch.float32)
    grads = {k: v.grad for k, v in st.items() if v.grad is not None}
    return logits.detach(), loss.detach(), grads, {k: v.detach() for k, v in st.items()}


class _View:
    """A ConnectomeBatch-like view of the same graphs in another presentation order."""

    def __init__(self, b, reverse_graphs: bool, reverse_edges: bool):
        ptr, B = b.ptr.cpu(), int(b.ptr.numel()) - 1
        order = list(range(B - 1, -1, -1)) if reverse_graphs else list(range(B))
        new_nodes = torch.cat([torch.arange(int(ptr[g]), int(ptr[g + 1])) for g in order]) if B else ptr[:0]
        inv = torch.empty_like(new_nodes)
        inv[new_nodes] = torch.arange(new_nodes.numel())
        ei = inv[b.edge_index.cpu()]
        ew = b.edge_weight.cpu()
        if reverse_edges:
            ei, ew = ei.flip(1), ew.flip(0)
        sizes = (ptr[1:] - ptr[:-1])[order]
        self.node_features = b.node_features.cpu()[new_nodes]
        self.edge_index, self.edge_weight = ei.contiguous(), ew.contiguous()
        self.batch = torch.repeat_interleave(torch.arange(B), sizes)
        self.labels = b.labels.cpu()[order]
        self.ptr = torch.cat([torch.zeros(1, dtype=torch.long), torch.cumsum(sizes, 0)])
        self.new_nodes, self.order = new_nodes, order

    def masks(self, masks: Optional[dict]) -> Optional[dict]:
        if masks is None:
            return None
        return {"layers": [m[self.new_nodes] for m in masks["layers"]], "head": masks["head"][self.order]}


class NoiseFloor:
    """Rule (3): lazily evaluated fp32 oracle gradients of the same batch in three other orders."""

    def __init__(self, kind, state_dict, b, dropout: float = 0.0, masks: Optional[dict] = None):
        self.args = (kind, {k: v.detach().cpu().clone() for k, v in state_dict.items()}, b, dropout, masks)
        self._grads = None

    def grads(self):
        if self._grads is None:
            kind, sd, b, dropout, masks = self.args
            self._grads = []
            for rg, re_ in ((True, False), (False, True), (True, True)):
                v = _View(b, rg, re_)
                self._grads.append(oracle_run(kind, sd, v, dropout, True, v.masks(masks))[2])
        return self._grads


def assert_grad(name: str, got: torch.Tensor, g32: torch.Tensor, g64: torch.Tensor, where: str = "",
                floor: Optional[NoiseFloor] = None) -> None:
    """Rules (1), (2), (3) of the module docstring, in that order; tallies the decision."""
    got = got.detach().cpu()
    ARBITER["checked"] += 1
    if float((got - g32).abs().max()) <= 1e-5 * float(g32.abs().max()) + 2e-6:
        return
    try:
        torch.testing.assert_close(got, g32, **GTOL)
        return
    except AssertionError:
        pass
    err_gpu = float((got.double() - g64).abs().max())
    err_cpu = float((g32.double() - g64).abs().max())
    ARBITER["fp64"] += 1
    ARBITER["names"].append(f"{where}:{name} gpu={err_gpu:.2e} cpu32={err_cpu:.2e} scale={float(g64.abs().max()):.2e}")
    if err_gpu <= err_cpu + 1e-9:
        return
    others = [float((g[name].double() - g64).abs().max()) for g in floor.grads()] if floor is not None else []
    ARBITER["floor"] += 1
    ARBITER["names"][-1] += " floor=" + "/".join(f"{e:.2e}" for e in others)
    assert err_gpu <= max([err_cpu] + others) + 1e-9, \
        f"{where}:{name}: HIP is {err_gpu:.2e} from the fp64 oracle; fp32 oracle evaluations: " \
        f"{err_cpu:.2e} (as given), {others} (other presentation orders)"


def recorded_masks(model, num_nodes: int, num_graphs: int) -> dict:
    """model.last_dropout (keep-bit bytes per layer + the head's factor) -> {0,1} float masks."""
    from connectome_gnn_amd import ops
    rec = model.last_dropout
    hid = model.batch_norms[0].num_features
    layers = [ops.unpack_keep_bits(m, num_nodes, hid).cpu() for m in rec["layers"]]
    # head factor = relu'(z) * keep / (1 - p): where z <= 0 the keep decision is immaterial
    head = (rec["head_factor"] > 0).float().cpu()
    assert head.shape[0] == num_graphs
    return {"layers": layers, "head": head}
