"""ORACLE -- test infrastructure only, never the product path.

CPU restatement (pure PyTorch, fp32, autograd) of the reference's batched
message-passing hot path.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this module; connectome_gnn_amd never
does, and it raises when its HIP library is missing instead of falling back.

Parity status: PINNED.  tests/test_oracle_golden.py checks every function
below against tests/golden/g1..g7 (*.npz), which tests/golden/make_goldens.py
recorded from the real reference (/root/reference, v0.2.0) in the build
container.

The restatement is functional (a flat ``state`` dict with the reference's
state_dict keys) rather than a copy of the reference's module tree, but it
issues the same ATen op sequence per layer, so it doubles as the timed
"port" CPU baseline (bench.py cpu_baseline.kind == "port").

Reference lines restated (relative to /root/reference/connectome_gnn/):
  scatter_sum / scatter_mean / graph_mean_pool   models.py:40-59
  gcn_layer                                      models.py:84-114
  sage_layer                                     models.py:136-152
  gcn_encode / gcn_forward                       models.py:203-216
  sage_encode / sage_forward                     models.py:256-266
  classifier head                                models.py:196-201
  BatchNorm1d semantics                          models.py:191-193 (torch)
  init_gcn_state / init_sage_state               models.py:78-82,130-134,176-201
  collate                                        graph.py:143-167
  train_epoch / evaluate / fit                   train.py:41-127
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

State = Dict[str, torch.Tensor]


# --------------------------------------------------------------------------
# containers + collate (graph.py:101-167)
# --------------------------------------------------------------------------
@dataclass
class OBatch:
    node_features: torch.Tensor   # [Nn, F]  f32
    edge_index: torch.Tensor      # [2, Ee]  i64, row0 = src, row1 = dst (global ids)
    edge_weight: torch.Tensor     # [Ee]     f32
    batch: torch.Tensor           # [Nn]     i64 graph id per node
    labels: Optional[torch.Tensor]
    ptr: torch.Tensor             # [B+1]    i64 cumulative node count

    @property
    def num_graphs(self) -> int:
        return int(self.ptr.numel()) - 1


def collate(node_feats: List[torch.Tensor], edge_indices: List[torch.Tensor],
            edge_weights: List[torch.Tensor], labels: Optional[List[torch.Tensor]]) -> OBatch:
    """graph.py:143-167 -- running node offset added to every graph's COO."""
    off, ptr, eis, bids = 0, [0], [], []
    for gi, (x, ei) in enumerate(zip(node_feats, edge_indices)):
        n = x.shape[0]
        eis.append(ei + off)                                    # :152
        bids.append(torch.full((n,), gi, dtype=torch.long))     # :154
        off += n
        ptr.append(off)                                         # :157-158
    return OBatch(torch.cat(node_feats, 0), torch.cat(eis, 1), torch.cat(edge_weights, 0),
                  torch.cat(bids, 0), torch.stack(labels) if labels else None,
                  torch.tensor(ptr, dtype=torch.long))


# --------------------------------------------------------------------------
# scatter helpers (models.py:40-59)
# --------------------------------------------------------------------------
def scatter_sum(src: torch.Tensor, index: torch.Tensor, dim_size: int) -> torch.Tensor:
    out = torch.zeros(dim_size, src.shape[1], dtype=src.dtype)
    return out.scatter_add_(0, index.unsqueeze(1).expand_as(src), src)


def scatter_mean(src: torch.Tensor, index: torch.Tensor, dim_size: int) -> torch.Tensor:
    tot = scatter_sum(src, index, dim_size)
    cnt = torch.zeros(dim_size, 1, dtype=src.dtype)
    cnt.scatter_add_(0, index.unsqueeze(1), torch.ones(index.shape[0], 1))
    return tot / (cnt + 1e-8)                                   # :47


def graph_mean_pool(x: torch.Tensor, batch: torch.Tensor, num_graphs: int) -> torch.Tensor:
    return scatter_mean(x, batch, num_graphs)                   # :57-59


# --------------------------------------------------------------------------
# layers
# --------------------------------------------------------------------------
def gcn_layer(x, edge_index, edge_weight, weight, bias):
    """models.py:84-114.  Self-loops (w=1) appended last; SOURCE-side degree."""
    n = x.shape[0]
    ar = torch.arange(n)
    s = torch.cat([edge_index[0], ar])                          # :98
    d = torch.cat([edge_index[1], ar])                          # :99
    w = torch.cat([edge_weight, torch.ones(n)])                 # :100
    deg = torch.zeros(n).scatter_add_(0, s, w)                  # :103-104
    dis = (deg + 1e-8).pow(-0.5)                                # :105
    coef = dis[s] * w * dis[d]                                  # :108
    t = F.linear(x, weight)                                     # :111
    msg = t[s] * coef.unsqueeze(1)                              # :112
    return scatter_sum(msg, d, n) + bias                        # :113-114


# ReLU decisions (test hook).  None: F.relu.  Else a callable (site, pre) -> {0,1} mask or None: the
# parity tests use it to LIST the pre-activations that sit within rounding of zero and to re-evaluate the
# path with such a tie decided the other way (tests/parity.py: a fp32 evaluation may land on either side
# of a tie, and one decision carries ~1/N of a gradient) -- the arithmetic is untouched.
RELU_HOOK = None


def _relu(pre, site: str):
    mask = RELU_HOOK(site, pre) if RELU_HOOK is not None else None
    return F.relu(pre) if mask is None else pre * mask.to(pre.dtype)


def sage_layer(x, edge_index, edge_weight, weight, bias, site: str = "sage"):
    """models.py:136-152.  Weighted mean over in-edges, concat, linear, ReLU."""
    n = x.shape[0]
    s, d = edge_index[0], edge_index[1]
    msg = x[s] * edge_weight.unsqueeze(1)                       # :146
    wsum = torch.zeros(n, 1).scatter_add_(0, d.unsqueeze(1), edge_weight.unsqueeze(1))  # :147-148
    agg = scatter_sum(msg, d, n) / (wsum + 1e-8)                # :149
    return _relu(F.linear(torch.cat([x, agg], 1), weight, bias), site)  # :151-152


def _bn(x, state: State, i: int, training: bool):
    """nn.BatchNorm1d(H): batch stats over ALL nodes in train, running stats in eval;
    momentum 0.1, eps 1e-5, running_var gets the unbiased variance."""
    p = f"batch_norms.{i}."
    if training:
        state[p + "num_batches_tracked"] += 1
    return F.batch_norm(x, state[p + "running_mean"], state[p + "running_var"],
                        state[p + "weight"], state[p + "bias"], training, 0.1, 1e-5)


def _num_layers(state: State) -> int:
    return 1 + max(int(k.split(".")[1]) for k in state if k.startswith("convs."))


def _dropout(x, p: float, training: bool, keep: Optional[torch.Tensor] = None):
    """F.dropout (models.py:199,210,261).  With ``keep`` (a {0,1} tensor shaped like x) the Bernoulli
    draw is replaced by the given decisions and the arithmetic is aten::native_dropout's:
    noise = keep / (1 - p), out = x * noise.  That is how the dropout-on mode is pinned: the HIP
    path's own keep bits are replayed here, so everything but the random draw is compared."""
    if keep is None:
        return F.dropout(x, p, training)
    if not training or p == 0.0:
        return x
    return x * (keep.to(x.dtype) / (1.0 - p))


def _keep(masks: Optional[dict], key: str, i: Optional[int] = None):
    if masks is None:
        return None
    return masks[key] if i is None else masks[key][i]


def _classifier(h, state: State, p: float, training: bool, keep=None):
    h = _relu(F.linear(h, state["classifier.0.weight"], state["classifier.0.bias"]), "head")
    h = _dropout(h, p, training, keep)
    return F.linear(h, state["classifier.3.weight"], state["classifier.3.bias"])  # :196-201


def gcn_encode(state: State, b: OBatch, dropout: float = 0.3, training: bool = False,
               masks: Optional[dict] = None):
    """masks (optional): {"layers": [keep [Nn,H]] per layer, "head": keep [B,H/2]}, see _dropout."""
    x = b.node_features
    for i in range(_num_layers(state)):                         # :206-210
        x = gcn_layer(x, b.edge_index, b.edge_weight,
                      state[f"convs.{i}.linear.weight"], state[f"convs.{i}.bias"])
        x = _bn(x, state, i, training)
        x = _relu(x, f"layer{i}")
        x = _dropout(x, dropout, training, _keep(masks, "layers", i))
    return graph_mean_pool(x, b.batch, b.num_graphs)            # :211


def gcn_forward(state: State, b: OBatch, dropout: float = 0.3, training: bool = False,
                masks: Optional[dict] = None):
    return _classifier(gcn_encode(state, b, dropout, training, masks), state, dropout, training,
                       _keep(masks, "head"))


def sage_encode(state: State, b: OBatch, dropout: float = 0.3, training: bool = False,
                masks: Optional[dict] = None):
    x = b.node_features
    for i in range(_num_layers(state)):                         # :258-261 (no ReLU after BN)
        x = sage_layer(x, b.edge_index, b.edge_weight,
                       state[f"convs.{i}.linear.weight"], state[f"convs.{i}.linear.bias"], f"layer{i}")
        x = _bn(x, state, i, training)
        x = _dropout(x, dropout, training, _keep(masks, "layers", i))
    return graph_mean_pool(x, b.batch, b.num_graphs)            # :262


def sage_forward(state: State, b: OBatch, dropout: float = 0.3, training: bool = False,
                 masks: Optional[dict] = None):
    return _classifier(sage_encode(state, b, dropout, training, masks), state, dropout, training,
                       _keep(masks, "head"))


FORWARD = {"gcn": gcn_forward, "sage": sage_forward}
ENCODE = {"gcn": gcn_encode, "sage": sage_encode}


# --------------------------------------------------------------------------
# parameter initialisation: consumes the torch RNG in the reference's order
# --------------------------------------------------------------------------
def _kaiming_linear(out_f: int, in_f: int, bias: bool):
    """nn.Linear.reset_parameters: kaiming_uniform(a=sqrt(5)) then bias U(-1/sqrt(fan_in), ..)."""
    w = torch.empty(out_f, in_f)
    bound = 1.0 / math.sqrt(in_f)       # gain*sqrt(3/fan_in) with gain = sqrt(2/(1+5)) = sqrt(1/3)
    w.uniform_(-bound, bound)
    b = torch.empty(out_f).uniform_(-bound, bound) if bias else None
    return w, b


def _xavier_(w: torch.Tensor):
    a = math.sqrt(6.0 / (w.shape[0] + w.shape[1]))
    return w.uniform_(-a, a)


def _init_common(state: State, hidden: int, num_classes: int, num_layers: int):
    for i in range(num_layers):
        state[f"batch_norms.{i}.weight"] = torch.ones(hidden)
        state[f"batch_norms.{i}.bias"] = torch.zeros(hidden)
        state[f"batch_norms.{i}.running_mean"] = torch.zeros(hidden)
        state[f"batch_norms.{i}.running_var"] = torch.ones(hidden)
        state[f"batch_norms.{i}.num_batches_tracked"] = torch.tensor(0, dtype=torch.long)
    w, b = _kaiming_linear(hidden // 2, hidden, True)
    state["classifier.0.weight"], state["classifier.0.bias"] = w, b
    w, b = _kaiming_linear(num_classes, hidden // 2, True)
    state["classifier.3.weight"], state["classifier.3.bias"] = w, b


def init_gcn_state(in_channels: int, hidden: int = 64, num_classes: int = 2,
                   num_layers: int = 3) -> State:
    """models.py:78-82 per layer (kaiming draw, then xavier redraw), :187-201."""
    st: State = {}
    dims = [in_channels] + [hidden] * num_layers
    for i in range(num_layers):
        w, _ = _kaiming_linear(dims[i + 1], dims[i], False)
        st[f"convs.{i}.bias"] = torch.zeros(dims[i + 1])
        st[f"convs.{i}.linear.weight"] = _xavier_(w)
    _init_common(st, hidden, num_classes, num_layers)
    return st


def init_sage_state(in_channels: int, hidden: int = 64, num_classes: int = 2,
                    num_layers: int = 3) -> State:
    """models.py:130-134 per layer, :241-254."""
    st: State = {}
    dims = [in_channels] + [hidden] * num_layers
    for i in range(num_layers):
        w, b = _kaiming_linear(dims[i + 1], 2 * dims[i], True)
        st[f"convs.{i}.linear.weight"] = _xavier_(w)
        st[f"convs.{i}.linear.bias"] = b
    _init_common(st, hidden, num_classes, num_layers)
    return st


INIT = {"gcn": init_gcn_state, "sage": init_sage_state}


def param_keys(state: State) -> List[str]:
    return [k for k in state if "running" not in k and "num_batches" not in k]


def require_grad(state: State) -> State:
    for k in param_keys(state):
        state[k] = state[k].detach().clone().requires_grad_(True)
    return state


# --------------------------------------------------------------------------
# training step / loop (train.py:41-127)
# --------------------------------------------------------------------------
def train_step(kind: str, state: State, b: OBatch, opt: torch.optim.Optimizer,
               dropout: float = 0.3) -> float:
    """train.py:46-52: zero_grad, forward, CE (mean over graphs), backward, step."""
    opt.zero_grad()
    logits = FORWARD[kind](state, b, dropout, True)
    loss = F.cross_entropy(logits, b.labels)
    loss.backward()
    opt.step()
    return float(loss.detach())


def train_epoch(kind, state, batches: List[OBatch], opt, dropout=0.3) -> float:
    tot, cnt = 0.0, 0
    for b in batches:
        tot += train_step(kind, state, b, opt, dropout) * b.num_graphs
        cnt += b.num_graphs
    return tot / max(cnt, 1)


@torch.no_grad()
def evaluate(kind, state, batches: List[OBatch]) -> dict:
    tot, correct, cnt = 0.0, 0, 0
    for b in batches:
        logits = FORWARD[kind](state, b, 0.0, False)
        tot += float(F.cross_entropy(logits, b.labels)) * b.num_graphs
        correct += int((logits.argmax(1) == b.labels).sum())
        cnt += b.num_graphs
    return {"accuracy": correct / max(cnt, 1), "loss": tot / max(cnt, 1),
            "correct": correct, "total": cnt}


def fit(kind, state, train_batches, val_batches, opt, num_epochs=50, patience=10,
        dropout=0.3) -> dict:
    hist = {"train_loss": [], "val_loss": [], "val_acc": []}
    best, best_ep, best_state = float("inf"), 0, None
    for ep in range(1, num_epochs + 1):
        tl = train_epoch(kind, state, train_batches, opt, dropout)
        ev = evaluate(kind, state, val_batches)
        hist["train_loss"].append(tl)
        hist["val_loss"].append(ev["loss"])
        hist["val_acc"].append(ev["accuracy"])
        if ev["loss"] < best:                                   # train.py:113-116
            best, best_ep = ev["loss"], ep
            best_state = {k: v.detach().clone() for k, v in state.items()}
        if ep - best_ep >= patience:
            break
    if best_state is not None:                                  # train.py:124-125
        with torch.no_grad():
            for k, v in best_state.items():
                state[k].copy_(v)
    return hist


# --------------------------------------------------------------------------
# dense fp64 cross-check of the layer algebra (independent formulation)
# --------------------------------------------------------------------------
def gcn_layer_dense64(x, edge_index, edge_weight, weight, bias):
    """Y = (D^-1/2 (A+I) D^-1/2)^T X W^T + b with row-sum (source-side) degree, fp64."""
    n = x.shape[0]
    a = torch.zeros(n, n, dtype=torch.float64)
    a.index_put_((edge_index[0], edge_index[1]), edge_weight.double(), accumulate=True)
    a += torch.eye(n, dtype=torch.float64)
    dis = (a.sum(1) + 1e-8).pow(-0.5)
    ahat = dis[:, None] * a * dis[None, :]
    return ahat.t() @ (x.double() @ weight.double().t()) + bias.double()


def sage_layer_dense64(x, edge_index, edge_weight, weight, bias):
    n = x.shape[0]
    a = torch.zeros(n, n, dtype=torch.float64)
    a.index_put_((edge_index[0], edge_index[1]), edge_weight.double(), accumulate=True)
    wsum = a.sum(0)                                   # per-destination weight sum
    agg = (a.t() @ x.double()) / (wsum[:, None] + 1e-8)
    return F.relu(torch.cat([x.double(), agg], 1) @ weight.double().t() + bias.double())
