"""Connectome graph containers, batching and the loader -- the drop-in boundary's data side.

Public names, fields, signatures and behaviour mirror the reference
(connectome_gnn/graph.py:27-197): ``ConnectomeGraph``, ``ConnectomeBatch`` (six positional
fields, ``.to``, ``.num_graphs``, ``.num_nodes``), ``collate_graphs``, ``ConnectomeDataLoader``
(``len == ceil(n / batch_size)``, last batch partial, ``torch.randperm`` on the global RNG).

What is new: a batch carries a lazily built, cached device-side *structure* (dst- and
src-sorted CSR, int32) that the HIP kernels consume; the int64 COO stays the public truth.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import torch


@dataclass
class ConnectomeGraph:
    """One subject: ``node_features [N,F]``, COO ``edge_index [2,E]`` (both directions stored),
    ``edge_weight [E]``, optional scalar ``label``.  (reference graph.py:27-94)"""

    node_features: torch.Tensor
    edge_index: torch.Tensor
    edge_weight: torch.Tensor
    label: Optional[torch.Tensor] = None
    subject_id: str = "unknown"

    @property
    def num_nodes(self) -> int:
        return self.node_features.shape[0]

    @property
    def num_edges(self) -> int:
        return self.edge_index.shape[1]

    @property
    def num_features(self) -> int:
        return self.node_features.shape[1]

    def adjacency_matrix(self) -> torch.Tensor:
        """Dense [N,N] weighted adjacency; a later duplicate edge overwrites (graph.py:72-78)."""
        n = self.num_nodes
        dense = self.edge_weight.new_zeros((n, n))
        dense[self.edge_index[0], self.edge_index[1]] = self.edge_weight
        return dense

    def degree(self) -> torch.Tensor:
        """Weighted out-degree per node (graph.py:80-85)."""
        out = self.edge_weight.new_zeros(self.num_nodes)
        return out.index_add_(0, self.edge_index[0], self.edge_weight)

    def to(self, device) -> "ConnectomeGraph":
        lab = None if self.label is None else self.label.to(device)
        return ConnectomeGraph(self.node_features.to(device), self.edge_index.to(device),
                               self.edge_weight.to(device), lab, self.subject_id)


@dataclass
class ConnectomeBatch:
    """Block-diagonal pack of B graphs (reference graph.py:101-140).

    node_features [Nn,F] f32 | edge_index [2,Ee] i64 (global node ids) | edge_weight [Ee] f32
    batch [Nn] i64 graph id per node | labels [B] or None | ptr [B+1] i64 cumulative nodes
    """

    node_features: torch.Tensor
    edge_index: torch.Tensor
    edge_weight: torch.Tensor
    batch: torch.Tensor
    labels: Optional[torch.Tensor]
    ptr: torch.Tensor
    _structure: object = field(default=None, init=False, repr=False, compare=False)
    _structure_key: object = field(default=None, init=False, repr=False, compare=False)
    # host int64 [B+1]: edges of graph g are the COO run [_eptr[g], _eptr[g+1]) -- set by
    # collate_graphs / the resident assembler, lets the structure be built per graph in LDS
    _eptr: object = field(default=None, init=False, repr=False, compare=False)

    @property
    def num_graphs(self) -> int:
        return int(self.ptr.shape[0]) - 1

    @property
    def num_nodes(self) -> int:
        return int(self.node_features.shape[0])

    def to(self, device) -> "ConnectomeBatch":
        lab = None if self.labels is None else self.labels.to(device)
        out = ConnectomeBatch(self.node_features.to(device), self.edge_index.to(device),
                              self.edge_weight.to(device), self.batch.to(device), lab,
                              self.ptr.to(device))
        # the cached structure stays valid if nothing moved
        if out.edge_index.data_ptr() == self.edge_index.data_ptr():
            out._structure, out._structure_key = self._structure, self._structure_key
        out._eptr = self._eptr
        return out

    def _edge_key(self):
        """Identity + in-place version of the edge tensors the cached structure was built from."""
        ei, ew = self.edge_index, self.edge_weight
        return (ei.data_ptr(), ei._version, tuple(ei.shape), ew.data_ptr(), ew._version)

    def invalidate(self) -> None:
        """Drop the cached device structure (call after changing edges through ``.data`` or other
        routes that bypass torch's version counter)."""
        self._structure = self._structure_key = None

    def structure(self):
        """Device CSR / blocked-ELL of this batch, built once by HIP kernels (structure.py) and
        cached.  The cache bakes in the edge list AND the edge weights; it is keyed on the edge
        tensors' storage and in-place version counters, so reassigning ``edge_index`` /
        ``edge_weight`` or mutating them in place (edge dropout, augmentation) rebuilds it --
        the reference re-derives everything on every call (models.py:90-108)."""
        key = self._edge_key()
        if self._structure is None or self._structure_key != key:
            from .structure import BatchStructure
            if self._structure is not None:
                self._eptr = None            # the per-graph edge offsets described the old COO
            self._structure = BatchStructure.build(self)
            self._structure_key = key
        return self._structure


def collate_graphs(graphs: Sequence[ConnectomeGraph]) -> ConnectomeBatch:
    """Pack graphs block-diagonally (reference graph.py:143-167): node ids of graph g are
    shifted by the number of nodes before it; ``labels`` is stacked from the graphs that
    carry one (None if none do); bit-exact int64 indexing."""
    sizes = [g.num_nodes for g in graphs]
    ptr = torch.zeros(len(graphs) + 1, dtype=torch.long)
    if sizes:
        ptr[1:] = torch.cumsum(torch.tensor(sizes, dtype=torch.long), 0)
    offsets = ptr[:-1].tolist()
    edge_index = torch.cat([g.edge_index + off for g, off in zip(graphs, offsets)], dim=1)
    batch_ids = torch.repeat_interleave(torch.arange(len(graphs), dtype=torch.long),
                                        torch.tensor(sizes, dtype=torch.long))
    labelled = [g.label for g in graphs if g.label is not None]
    out = ConnectomeBatch(
        node_features=torch.cat([g.node_features for g in graphs], dim=0),
        edge_index=edge_index,
        edge_weight=torch.cat([g.edge_weight for g in graphs], dim=0),
        batch=batch_ids,
        labels=torch.stack(labelled) if labelled else None,
        ptr=ptr,
    )
    eptr = torch.zeros(len(graphs) + 1, dtype=torch.long)
    if graphs:
        eptr[1:] = torch.cumsum(torch.tensor([g.num_edges for g in graphs], dtype=torch.long), 0)
    out._eptr = eptr
    return out


class ConnectomeDataLoader:
    """Minimal loader (reference graph.py:174-197): ``ceil(n/bs)`` batches, the last one
    partial, optional shuffle drawn from torch's global RNG once per epoch.

    ``rank``/``world_size`` (new, default single process) make every rank iterate the same
    global batches and keep its contiguous 1/world_size run of each -- the graph-sharded
    data-parallel layout of SURVEY 8e.  The global permutation is identical on all ranks
    as long as they seed torch identically.
    """

    def __init__(self, dataset: List[ConnectomeGraph], batch_size: int = 16, shuffle: bool = True,
                 rank: int = 0, world_size: int = 1):
        if world_size < 1 or not (0 <= rank < world_size):
            raise ValueError(f"bad rank/world_size {rank}/{world_size}")
        self.dataset = dataset
        self.batch_size = batch_size
        self.shuffle = shuffle
        self.rank = rank
        self.world_size = world_size

    def __len__(self) -> int:
        return -(-len(self.dataset) // self.batch_size)

    def __iter__(self):
        n = len(self.dataset)
        order = torch.randperm(n).tolist() if self.shuffle else list(range(n))
        for lo in range(0, n, self.batch_size):
            chunk = order[lo:lo + self.batch_size]
            if self.world_size > 1:
                if len(chunk) < self.world_size:
                    continue          # a tail smaller than the world: dropped on every rank
                chunk = shard_slice(chunk, self.rank, self.world_size)
            yield collate_graphs([self.dataset[i] for i in chunk])


def shard_slice(items: list, rank: int, world_size: int) -> list:
    """Contiguous shard of a global batch: sizes differ by at most one, earlier ranks get the
    extra element; concatenating all shards in rank order restores ``items``."""
    n = len(items)
    base, extra = divmod(n, world_size)
    lo = rank * base + min(rank, extra)
    return items[lo:lo + base + (1 if rank < extra else 0)]
