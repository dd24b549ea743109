#!/usr/bin/env python3
"""Generate the golden vectors G1-G8 of SURVEY.md section 8c.

Runs ONLY in the build container: it imports the real reference from
/root/reference (read-only) and records inputs + outputs of the hot path
(connectome_gnn/models.py:40-266, graph.py:143-167, train.py:41-127) as
plain-array .npz files (no pickle).  The reference itself never travels to
the GPU box; the fixtures do.

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python3 /root/repo/tests/golden/make_goldens.py

Key naming inside every .npz: '<group>__<name>' ('.' of state_dict keys kept).
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

REF = os.environ.get("CGNN_REFERENCE", "/root/reference")
sys.path.insert(0, REF)
sys.dont_write_bytecode = True

import connectome_gnn  # noqa: E402  (the reference)
from connectome_gnn.graph import ConnectomeGraph, ConnectomeDataLoader, collate_graphs  # noqa: E402
from connectome_gnn.models import (GCNConnectome, GCNLayer, GraphSAGEConnectome,  # noqa: E402
                                   SAGELayer)
from connectome_gnn.synthetic import generate_connectome, generate_dataset  # noqa: E402
from connectome_gnn.train import Trainer  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
torch.set_num_threads(1)  # deterministic reduction order for the record


def npy(t):
    return t.detach().cpu().numpy()


def batch_arrays(b, prefix="batch"):
    d = {
        f"{prefix}__node_features": npy(b.node_features),
        f"{prefix}__edge_index": npy(b.edge_index),
        f"{prefix}__edge_weight": npy(b.edge_weight),
        f"{prefix}__batch": npy(b.batch),
        f"{prefix}__ptr": npy(b.ptr),
    }
    if b.labels is not None:
        d[f"{prefix}__labels"] = npy(b.labels)
    return d


def graph_edge_counts(graphs):
    return np.asarray([g.num_edges for g in graphs], dtype=np.int64)


def model_record(cls, name, batch, hidden, seed=42):
    """G3-style record: init state, eval outputs, one train step (dropout 0)."""
    d = {}
    torch.manual_seed(seed)
    m = cls(batch.node_features.shape[1], hidden)
    for k, v in m.state_dict().items():
        d[f"{name}_init__{k}"] = npy(v)
    # also pin that a non-default dropout does not change the init stream
    m.eval()
    with torch.no_grad():
        d[f"{name}_eval__logits"] = npy(m(batch))
        d[f"{name}_eval__encode"] = npy(m.encode(batch))
    # train-mode forward/backward with dropout disabled (BN batch statistics)
    torch.manual_seed(seed)
    m = cls(batch.node_features.shape[1], hidden, dropout=0.0)
    m.train()
    logits = m(batch)
    loss = torch.nn.CrossEntropyLoss()(logits, batch.labels)
    loss.backward()
    d[f"{name}_train__logits"] = npy(logits)
    d[f"{name}_train__loss"] = npy(loss)
    for k, p in m.named_parameters():
        d[f"{name}_grad__{k}"] = npy(p.grad)
    for k, v in m.state_dict().items():
        if "running" in k or "num_batches" in k:
            d[f"{name}_after__{k}"] = npy(v)
    return d


def g1():
    graphs = generate_dataset(8, num_regions=20, seed=0)
    b = collate_graphs(graphs)
    d = batch_arrays(b)
    d["edge_counts"] = graph_edge_counts(graphs)
    np.savez_compressed(os.path.join(OUT, "g1_collate_8x20.npz"), **d)
    return b


def g2():
    # 6 nodes: node 4 isolated, node 5 has out-edges only (no in-edges),
    # edge (0->1) duplicated with different weights, asymmetric weights.
    src = torch.tensor([0, 0, 1, 2, 3, 5, 5, 2, 0], dtype=torch.long)
    dst = torch.tensor([1, 1, 2, 0, 2, 0, 3, 3, 3], dtype=torch.long)
    w = torch.tensor([0.5, 0.25, 1.5, 0.75, 0.3, 0.9, 0.2, 0.6, 1.1], dtype=torch.float32)
    ei = torch.stack([src, dst])
    torch.manual_seed(3)
    x = torch.randn(6, 4)
    cot = torch.randn(6, 7)
    d = {"x": npy(x), "edge_index": npy(ei), "edge_weight": npy(w), "cotangent": npy(cot)}
    for name, cls in (("gcn", GCNLayer), ("sage", SAGELayer)):
        torch.manual_seed(11)
        layer = cls(4, 7)
        with torch.no_grad():
            if name == "gcn":
                layer.bias.copy_(torch.linspace(-0.3, 0.3, 7))
        for k, v in layer.state_dict().items():
            d[f"{name}_param__{k}"] = npy(v)
        for tag, c in (("sum", torch.ones(6, 7)), ("cot", cot)):
            xi = x.clone().requires_grad_(True)
            layer.zero_grad()
            out = layer(xi, ei, w)
            (out * c).sum().backward()
            d[f"{name}_{tag}__out"] = npy(out)
            d[f"{name}_{tag}__dx"] = npy(xi.grad)
            for k, p in layer.named_parameters():
                d[f"{name}_{tag}__d_{k}"] = npy(p.grad)
    np.savez_compressed(os.path.join(OUT, "g2_layers_6node.npz"), **d)


def g3(b):
    d = batch_arrays(b)
    d.update(model_record(GCNConnectome, "gcn", b, 32))
    d.update(model_record(GraphSAGEConnectome, "sage", b, 32))
    np.savez_compressed(os.path.join(OUT, "g3_models_8x20_h32.npz"), **d)


def g4():
    graphs = generate_dataset(4, num_regions=84, k=8, seed=1)
    b = collate_graphs(graphs)
    d = batch_arrays(b)
    d.update(model_record(GCNConnectome, "gcn", b, 64))
    d.update(model_record(GraphSAGEConnectome, "sage", b, 64))
    np.savez_compressed(os.path.join(OUT, "g4_models_4x84_h64.npz"), **d)


def g5():
    graphs = generate_dataset(2, num_regions=360, k=14, seed=2)
    b = collate_graphs(graphs)
    d = batch_arrays(b)
    d.update(model_record(GCNConnectome, "gcn", b, 64))
    d.update(model_record(GraphSAGEConnectome, "sage", b, 128))
    np.savez_compressed(os.path.join(OUT, "g5_models_2x360.npz"), **d)


def g6():
    graphs = [
        generate_connectome(num_regions=20, k=4, seed=5),
        generate_connectome(num_regions=35, k=6, seed=6),
        generate_connectome(num_regions=84, k=8, seed=7),
    ]
    b = collate_graphs(graphs)
    d = batch_arrays(b)
    d["edge_counts"] = graph_edge_counts(graphs)
    d.update(model_record(GCNConnectome, "gcn", b, 32))
    d.update(model_record(GraphSAGEConnectome, "sage", b, 32))
    np.savez_compressed(os.path.join(OUT, "g6_mixed_20_35_84.npz"), **d)


def g7():
    graphs = generate_dataset(40, num_regions=20, seed=7)
    allb = collate_graphs(graphs)
    d = batch_arrays(allb, "all")
    d["edge_counts"] = graph_edge_counts(graphs)
    for name, cls in (("gcn", GCNConnectome), ("sage", GraphSAGEConnectome)):
        torch.manual_seed(42)
        m = cls(5, 32, dropout=0.0)
        opt = torch.optim.Adam(m.parameters(), lr=1e-3)
        tr = Trainer(m, opt, device="cpu")
        hist = tr.fit(ConnectomeDataLoader(graphs[:30], batch_size=10, shuffle=False),
                      ConnectomeDataLoader(graphs[30:], batch_size=10, shuffle=False),
                      num_epochs=3, patience=8, verbose=False)
        for k, v in hist.items():
            d[f"{name}_hist__{k}"] = np.asarray(v, dtype=np.float64)
        for k, v in m.state_dict().items():
            d[f"{name}_final__{k}"] = npy(v)
        ev = tr.evaluate(ConnectomeDataLoader(graphs[30:], batch_size=10, shuffle=False))
        d[f"{name}_eval__accuracy"] = np.float64(ev["accuracy"])
        d[f"{name}_eval__loss"] = np.float64(ev["loss"])
        d[f"{name}_eval__correct"] = np.int64(ev["correct"])
        d[f"{name}_eval__total"] = np.int64(ev["total"])
    np.savez_compressed(os.path.join(OUT, "g7_trainer_40x20.npz"), **d)


def g8():
    g = generate_connectome(seed=42)
    d = {
        "node_features": npy(g.node_features),
        "edge_index": npy(g.edge_index),
        "edge_weight": npy(g.edge_weight),
        "label": npy(g.label),
    }
    np.savez_compressed(os.path.join(OUT, "g8_generate_seed42.npz"), **d)


def main():
    print("reference version", connectome_gnn.__version__, "torch", torch.__version__,
          "numpy", np.__version__)
    b = g1()
    g2()
    g3(b)
    g4()
    g5()
    g6()
    g7()
    g8()
    tot = 0
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            s = os.path.getsize(os.path.join(OUT, f))
            tot += s
            print(f"{f:32s} {s:8d} B")
    print("total", tot)


if __name__ == "__main__":
    main()
