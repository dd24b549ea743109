"""Device-resident dataset and on-device batch assembly.

The reference collates on the host in a per-graph Python loop (graph.py:149-158, 0.19 s per
512 graphs) and copies six tensors per step (train.py:46; 454 MB at 4096 x 360 ROI).  For
regular datasets (every graph n nodes, e edges -- true of the Watts-Strogatz connectomes) the
whole dataset sits in HBM as dense arrays and a batch is a gather by subject id plus the
block-diagonal node offset: same int64 content as ``collate_graphs``, no host work.
"""
from __future__ import annotations

import torch

from .graph import ConnectomeBatch, shard_slice
from .synthetic import PackedDataset


def assemble_batch(ds: PackedDataset, subject_ids: torch.Tensor) -> ConnectomeBatch:
    """Bit-identical to ``collate_graphs([ds.graph(i) for i in subject_ids])``."""
    dev = ds.x.device
    ids = subject_ids.to(dev)
    b = int(ids.numel())
    n, e = int(ds.x.shape[1]), int(ds.edge_local.shape[2])
    offs = torch.arange(b, device=dev, dtype=torch.long) * n
    edge_index = (ds.edge_local.index_select(0, ids) + offs.view(b, 1, 1)).permute(1, 0, 2)
    out = ConnectomeBatch(
        node_features=ds.x.index_select(0, ids).reshape(b * n, -1),
        edge_index=edge_index.reshape(2, b * e).contiguous(),
        edge_weight=ds.edge_weight.index_select(0, ids).reshape(b * e),
        batch=torch.arange(b, device=dev, dtype=torch.long).repeat_interleave(n),
        labels=ds.labels.index_select(0, ids),
        ptr=torch.arange(b + 1, device=dev, dtype=torch.long) * n,
    )
    out._eptr = torch.arange(b + 1, dtype=torch.long) * e          # host: e edges per graph
    return out


class ResidentDataLoader:
    """``ConnectomeDataLoader`` semantics (len = ceil, partial last batch, global-RNG shuffle)
    over a PackedDataset that already lives on the device.  rank/world_size select this
    rank's contiguous shard of every global batch (graph-sharded data parallelism)."""

    def __init__(self, dataset: PackedDataset, batch_size: int = 16, shuffle: bool = True,
                 rank: int = 0, world_size: int = 1):
        self.dataset, self.batch_size, self.shuffle = dataset, batch_size, shuffle
        self.rank, self.world_size = rank, world_size

    def __len__(self) -> int:
        return -(-self.dataset.num_subjects // self.batch_size)

    def __iter__(self):
        n = self.dataset.num_subjects
        order = torch.randperm(n) if self.shuffle else torch.arange(n)
        for lo in range(0, n, self.batch_size):
            chunk = order[lo:lo + self.batch_size]
            if self.world_size > 1:
                if chunk.numel() < self.world_size:
                    continue          # a tail smaller than the world: dropped on every rank
                chunk = torch.tensor(shard_slice(chunk.tolist(), self.rank, self.world_size),
                                     dtype=torch.long)
            yield assemble_batch(self.dataset, chunk)
