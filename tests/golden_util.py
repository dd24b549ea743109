"""Helpers to read the golden fixtures (tests/golden/*.npz, written by make_goldens.py)."""
import os

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name: str) -> dict:
    with np.load(os.path.join(GOLDEN_DIR, name), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def group(d: dict, prefix: str) -> dict:
    """All entries '<prefix>__<key>' as torch tensors keyed by <key>."""
    p = prefix + "__"
    return {k[len(p):]: torch.from_numpy(np.asarray(v)) for k, v in d.items() if k.startswith(p)}


def split_graphs(bd: dict, edge_counts=None):
    """Undo collate: per-graph (x, local edge_index, edge_weight, label) from batch arrays."""
    ptr = bd["ptr"].tolist()
    ei, ew, x = bd["edge_index"], bd["edge_weight"], bd["node_features"]
    labels = bd.get("labels")
    out, e0 = [], 0
    for g in range(len(ptr) - 1):
        lo, hi = ptr[g], ptr[g + 1]
        if edge_counts is not None:
            e1 = e0 + int(edge_counts[g])
        else:
            m = (ei[0] >= lo) & (ei[0] < hi)
            e1 = e0 + int(m.sum())
        out.append((x[lo:hi].clone(), ei[:, e0:e1] - lo, ew[e0:e1].clone(),
                    labels[g] if labels is not None else None))
        e0 = e1
    return out
