"""connectome_gnn_amd -- the batched message-passing training path of connectome-gnn-suite,
rebuilt for AMD Instinct MI355X (gfx950): hand-written HIP kernels behind the reference's own
Python API.  The package exposes the reference's ten public names
(connectome_gnn/__init__.py:29-40), so ``import connectome_gnn_amd as connectome_gnn`` is the
whole migration for a script that trains on a ROCm device; the MI355X-specific extras
(``resident``, ``graphed``, ``dist``, ``ops``, ``optim``) are reached as submodules.
"""
from . import graph as _graph
from . import models as _models
from . import synthetic as _synthetic
from . import train as _train

__version__ = "0.2.0+mi355x.r1"

# public name -> defining module, in the reference's export order
_PUBLIC = (
    (_graph, ("ConnectomeGraph", "ConnectomeBatch", "ConnectomeDataLoader", "collate_graphs")),
    (_synthetic, ("generate_connectome", "generate_dataset", "REGION_NAMES")),
    (_models, ("GCNConnectome", "GraphSAGEConnectome")),
    (_train, ("Trainer",)),
)
__all__ = []
for _module, _names in _PUBLIC:
    for _name in _names:
        globals()[_name] = getattr(_module, _name)
        __all__.append(_name)
del _module, _names, _name
