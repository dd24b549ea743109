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
  (4) RELU TIES (assert_grads; used by the shape sweeps, where 10^6 ReLU decisions per run make a tie
      likely): if (1)-(3) fail, the float64 oracle's pre-activations within 2e-5 of their channel's
      median magnitude of zero are listed -- decisions an fp32 evaluation can take either way, each
      carrying ~1/N of a gradient -- and the oracle is re-evaluated with up to 4 of them decided the other
      way (greedily: the one that brings it closest to the HIP gradients first); the HIP path must then
      pass (1)-(3) with ALL its gradients against that evaluation.  Each use is printed with the ties it
      needed.
Every decision is tallied in ARBITER; tests/test_zz_arbiter.py prints the tally and fails if rules
(2)+(3) decided more than 5 % of the tensors, so they cannot silently become the norm.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from oracle import reference_path as O

TOL = dict(rtol=1e-5, atol=2e-6)
GTOL = dict(rtol=1e-4, atol=2e-6)
ARBITER = {"checked": 0, "fp64": 0, "floor": 0, "names": [], "relu_ties": []}


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


# ------------------------------------------------------------------ ReLU decisions within rounding of a tie
RELU_TIE_TOL = 2e-5        # |pre-activation| <= this x the channel's median magnitude = "a tie in fp32": a
                           # pre-activation is a sum of K ~ 10..512 products whose magnitudes add up to ~10x
                           # the channel's typical |pre|; the split-bf16 products truncate (one-sided, <= 2^-23
                           # each), fp32 CPU sums round per addition -- evaluations differ by up to ~1e-5 of it
RELU_TIE_MAX = 4           # at most this many ties are decided the other way
RELU_TIE_LIST = 64         # ... out of at most this many candidates (else the rule does not apply)


def relu_ties(kind: str, state_dict, b, dropout: float = 0.0, masks: Optional[dict] = None) -> list:
    """(site, row, channel) of every ReLU pre-activation of the float64 oracle that an fp32 evaluation may
    decide either way: |pre| <= RELU_TIE_TOL x the channel's median |pre| (the pre-activation is a sum of
    ~K products of that magnitude; fp32 sums of it differ by ~1e-7 of it between summation orders)."""
    found = []

    def hook(site, pre):
        p = pre.detach()
        scale = p.abs().median(dim=0).values.clamp_min(1e-30)
        for r, c in (p.abs() <= RELU_TIE_TOL * scale).nonzero().tolist():
            found.append((site, r, c))
        return None

    O.RELU_HOOK = hook
    try:
        oracle_run(kind, state_dict, b, dropout, True, masks, dtype=torch.float64)
    finally:
        O.RELU_HOOK = None
    return found


def oracle_run_flipped(kind: str, state_dict, b, flips, dropout: float = 0.0, masks: Optional[dict] = None,
                       dtype=torch.float32):
    """oracle_run with the ReLU decisions listed in ``flips`` taken the other way."""
    by_site = {}
    for site, r, c in flips:
        by_site.setdefault(site, []).append((r, c))

    def hook(site, pre):
        if site not in by_site:
            return None
        mask = pre.detach() > 0
        for r, c in by_site[site]:
            mask[r, c] = ~mask[r, c]
        return mask

    O.RELU_HOOK = hook
    try:
        return oracle_run(kind, state_dict, b, dropout, True, masks, dtype=dtype)
    finally:
        O.RELU_HOOK = None


def assert_grads(named_grads: Dict[str, torch.Tensor], kind: str, state_dict, b, where: str, dropout: float = 0.0,
                 masks: Optional[dict] = None, g32=None, g64=None) -> None:
    """Every gradient against the oracle by rules (1)-(3) (assert_grad); if some fail, rule (4): the oracle is
    re-evaluated with its ReLU ties (relu_ties) resolved the other way, subset by subset, and the HIP path
    must pass rules (1)-(3) against ONE of those evaluations with ALL its gradients.  Tallied and printed like
    the arbiter's decisions."""
    import itertools
    if g32 is None:
        _, _, g32, _ = oracle_run(kind, state_dict, b, dropout, True, masks)
        _, _, g64, _ = oracle_run(kind, state_dict, b, dropout, True, masks, dtype=torch.float64)
    floor = NoiseFloor(kind, state_dict, b, dropout, masks)

    def check(a32, a64, fl):
        snap = (ARBITER["checked"], ARBITER["fp64"], ARBITER["floor"], len(ARBITER["names"]))
        try:
            for k_, g in named_grads.items():
                assert_grad(k_, g, a32[k_], a64[k_], where, fl)
            return None
        except AssertionError as exc:                 # a failed attempt leaves no trace in the tally
            ARBITER["checked"], ARBITER["fp64"], ARBITER["floor"] = snap[0], snap[1], snap[2]
            del ARBITER["names"][snap[3]:]
            return exc

    first = check(g32, g64, floor)
    if first is None:
        return
    ties = relu_ties(kind, state_dict, b, dropout, masks)
    if not ties or len(ties) > RELU_TIE_LIST:
        raise AssertionError(f"{first} [ReLU ties within {RELU_TIE_TOL:g} of zero: {len(ties)}]")
    got = {k_: g.detach().cpu().double() for k_, g in named_grads.items()}

    def worst(a64):
        return max(float((got[k_] - a64[k_]).abs().max()) / max(float(a64[k_].abs().max()), 1e-30) for k_ in got)

    # greedy: decide the other way, one at a time, the tie that brings the float64 oracle closest to the HIP
    # gradients; after each, the full rules (1)-(3) against that evaluation (fp32 and fp64)
    chosen, best = [], worst(g64)
    for _ in range(RELU_TIE_MAX):
        trials = []
        for t in ties:
            if t not in chosen:
                _, _, a64, _ = oracle_run_flipped(kind, state_dict, b, chosen + [t], dropout, masks, dtype=torch.float64)
                trials.append((worst(a64), t, a64))
        w, t, a64 = min(trials, key=lambda v: v[0])
        if not w < best:
            break
        chosen.append(t)
        best = w
        _, _, a32, _ = oracle_run_flipped(kind, state_dict, b, chosen, dropout, masks)
        if check(a32, a64, None) is None:
            ARBITER["relu_ties"].append(f"{where}: passes with {chosen} decided the other way "
                                        f"({len(ties)} pre-activation(s) within {RELU_TIE_TOL:g} of zero)")
            return
    raise AssertionError(f"{first} [no resolution of the ReLU ties explains it: {len(ties)} candidates, tried {chosen}]")


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
