"""connectome_gnn_amd -- the batched message-passing training path of connectome-gnn-suite,
rebuilt for AMD Instinct MI355X (gfx950): hand-written HIP kernels behind the reference's own
Python API.  Exports the same ten names as the reference package
(connectome_gnn/__init__.py:29-40), so ``import connectome_gnn_amd as connectome_gnn`` is the
whole migration for a script that trains on a ROCm device.
"""
__version__ = "0.2.0+mi355x.r1"

from .graph import ConnectomeGraph, ConnectomeBatch, ConnectomeDataLoader, collate_graphs
from .synthetic import generate_connectome, generate_dataset, REGION_NAMES
from .models import GCNConnectome, GraphSAGEConnectome
from .train import Trainer

__all__ = [
    "ConnectomeGraph", "ConnectomeBatch", "ConnectomeDataLoader", "collate_graphs",
    "generate_connectome", "generate_dataset", "REGION_NAMES",
    "GCNConnectome", "GraphSAGEConnectome", "Trainer",
]
