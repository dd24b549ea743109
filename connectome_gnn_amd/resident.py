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
from .structure import call_prepare
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

    def __init__(self, dataset: PackedDataset, batch_size: int = 16, shuffle=True,
                 rank: int = 0, world_size: int = 1, prefetch: bool = False, prepare=None,
                 cache_batches: bool = False, structure_cache: bool = False):
        """shuffle: True (new random composition of every batch each epoch, the reference's
        semantics), False, or "batches": the batches are composed once (one random permutation)
        and only their ORDER is re-drawn every epoch.  cache_batches (needs shuffle False or
        "batches"): the assembled device batches -- with their CSR / blocked-ELL structure -- are
        kept and handed out again each epoch, so per-batch work is paid once and
        ``Trainer(graph=True)`` can replay one captured HIP graph per batch.

        structure_cache: build every SUBJECT's blocked-ELL / `dis` once (structure_cache.py) and
        assemble a batch's structure by three small gathers -- for the per-tile fused GCN encoder on
        datasets with one graph per tile (193..384 nodes); every epoch can then re-draw the batch
        composition (shuffle=True) at no per-batch build cost.

        prefetch: assemble the NEXT batch and build its structure (CSR, and whatever
        ``prepare(batch)`` builds, e.g. ``model.prepare_batch``) on a side stream while the caller
        trains on the current one -- the builds read sizes back to the host, and on the training
        stream those read-backs would wait for the whole previous step."""
        self.dataset, self.batch_size, self.shuffle = dataset, batch_size, shuffle
        self.rank, self.world_size = rank, world_size
        self.prefetch, self.prepare = prefetch, prepare
        self._side = None
        if cache_batches and shuffle is True:
            raise ValueError("cache_batches needs shuffle=False or shuffle='batches' (fixed batch composition)")
        self.cache_batches = cache_batches
        self._fixed_order = None           # composition of the batches for shuffle='batches'
        self._cache = None
        self.structure_cache = None
        if structure_cache:
            from .structure_cache import SubjectStructureCache
            self.structure_cache = SubjectStructureCache(dataset)

    def __len__(self) -> int:
        return -(-self.dataset.num_subjects // self.batch_size)

    def _chunks(self):
        """Subject ids of this rank's batches.  The permutation is drawn on the host (the global CPU
        generator, as the reference's loader does) and uploaded ONCE per epoch; a batch's ids are a
        slice of that device array, so handing a batch over costs no host-to-device copy."""
        n = self.dataset.num_subjects
        if self.shuffle == "batches":
            if self._fixed_order is None:
                self._fixed_order = torch.randperm(n)
            order = self._fixed_order
        else:
            order = torch.randperm(n) if self.shuffle else torch.arange(n)
        dev = self.dataset.x.device
        if dev.type == "cuda":
            # (from pinned memory, asynchronously: a pageable upload would make the host wait for every
            # step already queued on this stream -- one pipeline bubble per epoch.  Two pinned staging
            # buffers used in turn, each guarded by the event of its last upload: no allocation per epoch)
            order = self._upload(order, dev)
        for lo in range(0, n, self.batch_size):
            hi = min(n, lo + self.batch_size)
            if self.world_size > 1:
                if hi - lo < self.world_size:
                    continue          # a tail smaller than the world: dropped on every rank
                run = shard_slice(range(lo, hi), self.rank, self.world_size)   # a contiguous run
                lo, hi = run.start, run.stop
            yield order[lo:hi]

    def _upload(self, order: torch.Tensor, dev) -> torch.Tensor:
        ring = self.__dict__.setdefault("_pinned", [])
        if len(ring) < 2 or ring[0][0].numel() != order.numel():
            ring[:] = [[torch.empty(order.numel(), dtype=order.dtype).pin_memory(), None] for _ in range(2)]
            self._pin_turn = 0
        slot = ring[self._pin_turn]
        self._pin_turn ^= 1
        if slot[1] is not None:
            slot[1].synchronize()                  # the upload that last read this buffer (two epochs ago)
        slot[0].copy_(order)
        out = slot[0].to(dev, non_blocking=True)
        slot[1] = torch.cuda.Event()
        slot[1].record(torch.cuda.current_stream(dev))
        return out

    def __iter__(self):
        if self.structure_cache is not None and not self.cache_batches:
            from .structure_cache import ResidentBatch

            def build(chunk):                       # three gathers; the structure is the cache's
                return ResidentBatch(self.structure_cache, chunk)
        else:
            def build(chunk):
                return assemble_batch(self.dataset, chunk)
        if self.cache_batches:
            if self._cache is None:
                self._cache = []
                for chunk in self._chunks():
                    b = assemble_batch(self.dataset, chunk)
                    if self.prepare is not None:
                        call_prepare(self.prepare, b)        # kept batches: amortised structure work pays
                    self._cache.append(b)
            order = torch.randperm(len(self._cache)).tolist() if self.shuffle == "batches" \
                else range(len(self._cache))
            for i in order:
                yield self._cache[i]
            return
        if not self.prefetch or self.dataset.x.device.type != "cuda":
            for chunk in self._chunks():
                b = build(chunk)
                if self.prepare is not None:
                    self.prepare(b)
                yield b
            return
        dev = self.dataset.x.device
        if self._side is None:
            self._side = torch.cuda.Stream(device=dev)
        side = self._side

        def make(chunk, after):
            # Everything of the batch is allocated, built and read back on the side stream.  Its
            # memory is only ever recycled by a later side-stream allocation, which by then has
            # waited (`after`) for the training stream to be done with the batch it came from.
            if after is not None:
                side.wait_event(after)
            with torch.cuda.stream(side):
                b = build(chunk)
                b.structure()
                if self.prepare is not None:
                    self.prepare(b)
                ev = side.record_event()
            return b, ev

        main = torch.cuda.current_stream(dev)
        chunks = list(self._chunks())
        nxt = make(chunks[0], main.record_event()) if chunks else None
        for i in range(len(chunks)):
            cur, ev = nxt
            main.wait_event(ev)
            done_prev = main.record_event()        # all training work enqueued before batch i
            nxt = None
            yield cur
            # the caller has now enqueued its step on batch i; build batch i+1 while it runs (the
            # side stream only waits for the steps BEFORE it, whose batches' memory it may reuse)
            if i + 1 < len(chunks):
                nxt = make(chunks[i + 1], done_prev)
