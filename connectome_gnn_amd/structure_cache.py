"""Per-subject structure cache for device-resident datasets.

Re-shuffling a resident dataset every epoch (the reference's loader semantics, graph.py:192-197)
makes every batch new, and building a batch's structure -- on-device collate of the int64 COO
(454 MB at 4096 x 360 ROI), the two CSR orderings (342 MB), the two blocked-ELL arrays (424 MB)
-- costs 1.5 ms against a 2.1 ms step.  None of it depends on the batch: a subject's graph is
fixed, and when a tile holds exactly one graph the tile's blocked-ELL entries, its block offsets
and its `dis` vector ARE the subject's.  So they are built once per subject (with the ordinary HIP
builders, over the dataset in chunks, tiles cut at every graph) and a batch's structure becomes
three small gathers:

    blk_off_{dst,src}[tile] = cache.blk_off[subject]      (absolute offsets into the cache arrays)
    dis[tile rows]          = cache.dis[subject]
    entries                 = the cache arrays themselves  (never copied)

The fused kernels index blocks as tile_blk[tile] + b and take a block's width from consecutive
offsets, so tile_blk gets a stride of (blocks + 1) and every subject's offset row carries its end
offset: no kernel change.  The public COO fields of the batch (`edge_index`, `edge_weight`,
`batch`) are assembled lazily, only if something reads them.

Graphs of up to 192 nodes (84-ROI atlases) would share a tile in the per-batch build (two to four
graphs per 384-row tile, whose 16-row blocks straddle graphs and therefore belong to no subject);
under the cache every graph gets a tile of its own.  The tile kernels then walk more, smaller tiles
(512 x 84 rows instead of 256 x 168 for BASELINE config 2) -- a few per cent of GPU time against
issuing ~45 launches per step from Python, which is what a freshly shuffled batch costs otherwise
(`Trainer(graph=True)` replays ONE captured step per batch size over such a loader).

Scope: the per-tile fused GCN path (hidden 64) and the GraphSAGE encoder (sage_path.py; its own ELL
family without self-loops and `den` per subject, built on first use) on regular datasets (<= 384 nodes
per graph); anything else keeps the ordinary per-batch build.
"""
from __future__ import annotations

import weakref
from typing import Optional

import torch

from . import _lib
from .graph import ConnectomeBatch
from .structure import FusedMeta
from .synthetic import PackedDataset

MAX_ROWS = 384


KINDS = {"gcn": 1.0, "sage": 0.0}        # family -> weight of the ELL's self-loop entry (models.py:97-100 / :146)


class _Family:
    """One model family's per-subject arrays: blocked-ELL entries of both orderings, each subject's
    block-offset rows (absolute offsets into the entry arrays) and its per-node normaliser (GCN:
    `dis`, models.py:97-105; GraphSAGE: `den` = in-weight sum + 1e-8, models.py:146-149)."""
    __slots__ = ("ent_dst", "ent_src", "blk_off_dst", "blk_off_src", "norm")


class SubjectStructureCache:
    """Blocked-ELL entries (both orderings), block offsets and the per-node normaliser of every subject
    of a dataset, per model family (GCN: self-loop entry + `dis`; GraphSAGE: no self-loop + `den`).  The
    GCN family is built with the cache, the GraphSAGE one on first use."""

    def __init__(self, ds: PackedDataset, chunk: int = 2048):
        n = int(ds.x.shape[1])
        if not (0 < n <= MAX_ROWS):
            raise ValueError(f"structure cache: a graph must fit one LDS tile (<= {MAX_ROWS} nodes), got {n}")
        self.dataset, self.n, self._chunk = ds, n, chunk
        self.nb = (n + 15) // 16                                     # 16-row blocks per graph
        self._fam = {}
        self._static = {}
        self.family("gcn")

    def family(self, kind: str) -> _Family:
        if kind in self._fam:
            return self._fam[kind]
        from .resident import assemble_batch
        lib = _lib.load()
        ds, n = self.dataset, self.n
        dev = ds.x.device
        S = ds.num_subjects
        grid = int(lib.cgnn_fused_grid())
        ents = {"dst": [], "src": []}
        offs = {"dst": [], "src": []}
        norm = []
        base = {"dst": 0, "src": 0}
        for lo in range(0, S, self._chunk):
            ids = torch.arange(lo, min(S, lo + self._chunk), device=dev)
            b = assemble_batch(ds, ids)
            s = b.structure()
            if not s.block_diagonal:
                raise ValueError("structure cache: a subject has edges outside its graph")
            m = s.fused_meta(n, grid, KINDS[kind])      # row cap = one graph: a tile per subject
            if int(m.tile_ptr.numel()) - 1 != ids.numel():
                raise ValueError("structure cache: tiles are not one graph each")
            per_node = s.gcn_dis(m) if kind == "gcn" else s.sage_norm(backward_coef=False).den
            norm.append(per_node.view(ids.numel(), n).clone())
            for name, blk_off, ent in (("dst", m.blk_off_dst, m.ent_dst), ("src", m.blk_off_src, m.ent_src)):
                used = int(blk_off[-1])                               # entries (8 bytes each)
                ents[name].append(ent[:used * 8].clone())
                # per subject: offsets of its nb blocks + its end offset, absolute in the cache array
                first = torch.arange(ids.numel(), device=dev) * self.nb
                rows = blk_off[(first.view(-1, 1) + torch.arange(self.nb + 1, device=dev).view(1, -1))]
                offs[name].append(rows.to(torch.int64) + base[name])
                base[name] += used
            del b, s, m
        for name in ("dst", "src"):
            if base[name] > 2 ** 31 - 17:
                raise ValueError("structure cache exceeds 2^31 entries; cache a smaller dataset")
        f = _Family()
        f.ent_dst, f.ent_src = torch.cat(ents["dst"]), torch.cat(ents["src"])
        f.blk_off_dst = torch.cat(offs["dst"]).to(torch.int32).contiguous()     # [S, nb + 1]
        f.blk_off_src = torch.cat(offs["src"]).to(torch.int32).contiguous()
        f.norm = torch.cat(norm).contiguous()                                  # [S, n]
        self._fam[kind] = f
        return f

    # the GCN family under its historical names
    ent_dst = property(lambda self: self.family("gcn").ent_dst)
    ent_src = property(lambda self: self.family("gcn").ent_src)
    blk_off_dst = property(lambda self: self.family("gcn").blk_off_dst)
    blk_off_src = property(lambda self: self.family("gcn").blk_off_src)
    dis = property(lambda self: self.family("gcn").norm)

    def static(self, b: int):
        """Arrays that depend on the batch size only."""
        if b not in self._static:
            dev = self.dataset.x.device
            tile_ptr = (torch.arange(b + 1, device=dev, dtype=torch.int32) * self.n).contiguous()
            tile_blk = (torch.arange(b + 1, device=dev, dtype=torch.int32) * (self.nb + 1)).contiguous()
            node_graph = torch.arange(b, device=dev, dtype=torch.int32).repeat_interleave(self.n).contiguous()
            self._static[b] = (tile_ptr, tile_blk, node_graph)
        return self._static[b]


class CachedStructure:
    """What the one-node encoders over LDS tiles (per-tile GCN; GraphSAGE) ask of a batch structure,
    assembled from the cache.  `family(kind)` -- (block offsets dst, src, per-node normaliser) of the
    batch -- comes from the batch's own single assembly launch when its owner provides one
    (ResidentBatch), else from three gathers."""
    cached_subjects = True

    def __init__(self, cache: SubjectStructureCache, ids, b: int, assembled=None):
        """ids: the subject ids on the cache's device, or a callable that returns them.
        assembled(kind): callable returning (blk_off_dst, blk_off_src, norm) of the batch."""
        self.cache, self._ids_src, self._assembled = cache, ids, assembled
        self.num_graphs, self.num_nodes = b, b * cache.n
        self.max_nodes_per_graph = cache.n
        self.block_diagonal = True
        tile_ptr, tile_blk, node_graph = cache.static(b)
        self.gptr, self.node_graph = tile_ptr, node_graph
        self._metas = {}

    @property
    def _ids(self) -> torch.Tensor:
        return self._ids_src() if callable(self._ids_src) else self._ids_src

    def _family(self, kind: str):
        """(FusedMeta, per-node normaliser) of this batch for one model family."""
        if kind not in self._metas:
            cache, b = self.cache, self.num_graphs
            fam = cache.family(kind)
            tile_ptr, tile_blk, _ = cache.static(b)
            if self._assembled is not None:
                od, os_, norm = self._assembled(kind)
            else:
                od = fam.blk_off_dst.index_select(0, self._ids).view(-1)
                os_ = fam.blk_off_src.index_select(0, self._ids).view(-1)
                norm = fam.norm.index_select(0, self._ids).view(-1)
            meta = FusedMeta(tile_ptr, tile_blk, cache.n, b * (cache.nb + 1), od, fam.ent_dst, os_, fam.ent_src, None)
            self._metas[kind] = (meta, norm)
        return self._metas[kind]

    # -- the interface fused.py / sage_path.py / models.py use
    def fused_meta(self, max_rows: int, num_workgroups: int, self_weight: float = 1.0) -> FusedMeta:
        kind = {v: k for k, v in KINDS.items()}.get(float(self_weight))
        if max_rows != MAX_ROWS or kind is None:
            raise ValueError("cached structure: tiles of 384 rows, GCN (self-loop 1) or GraphSAGE (none) metadata")
        return self._family(kind)[0]

    def gcn_dis(self, meta: FusedMeta) -> torch.Tensor:
        """Per-subject `dis`, gathered (a function of the subject's edge weights only)."""
        return self._family("gcn")[1]

    def sage_norm(self, backward_coef: bool = True):
        """Per-subject `den`, gathered; the per-edge arrays of the gather-form kernels do not exist here
        (the tiled aggregate divides by `den` itself)."""
        from .structure import SageNorm
        return SageNorm(self._family("sage")[1], None, None)

    def tiles_struct(self, meta: FusedMeta, dis: Optional[torch.Tensor] = None):
        t = _lib.CgnnTiles()
        t.num_nodes = self.num_nodes
        t.num_tiles = self.num_graphs
        t.max_tile_rows = meta.max_tile_rows
        t.tile_ptr, t.tile_blk = meta.tile_ptr.data_ptr(), meta.tile_blk.data_ptr()
        t.blk_off_dst, t.ent_dst = meta.blk_off_dst.data_ptr(), meta.ent_dst.data_ptr()
        t.blk_off_src, t.ent_src = meta.blk_off_src.data_ptr(), meta.ent_src.data_ptr()
        t.dis = dis.data_ptr() if dis is not None else None
        return t

    def tiled_ok(self, width: int) -> bool:
        return width % 64 == 0 and self.num_nodes > 0

    def __getattr__(self, name):
        raise AttributeError(f"CachedStructure has no '{name}': it serves the per-tile fused GCN encoder and the "
                             "GraphSAGE encoder only (no CSR: use a loader without structure_cache for other paths)")


class ResidentBatch(ConnectomeBatch):
    """A ConnectomeBatch of a resident dataset whose structure comes from the subject cache.  Every
    field is assembled on first access -- node features, labels and the structure by gathers, the COO
    fields bit-identical to ``assemble_batch`` -- so handing the batch to a captured step that reads
    only its subject ids launches nothing."""

    def __init__(self, cache: SubjectStructureCache, ids: torch.Tensor, ids_offset: Optional[torch.Tensor] = None):
        """ids_offset (device int64 [1], optional): the batch is ``ids[offset : offset + len(ids)]`` of a
        longer id buffer that starts at ``ids`` -- the offset is read by the assembling kernel, so a
        captured step walks an epoch's permutation without any per-step host copy."""
        self._cache, self._ids_src = cache, ids      # as handed in (host ids of a loader: no copy yet)
        self._ids_offset = ids_offset
        self._b = int(ids.numel())
        self._lazy = {}
        self._coo = None
        # (the structure reaches back to its batch through a WEAK reference: a strong one would make every
        # batch a reference cycle that only the cyclic collector frees -- tens of MB of device tensors per
        # batch piling up until a generation-2 collection stalls the loop for ~100 ms)
        me = weakref.ref(self)
        self._structure = CachedStructure(cache, lambda: me()._ids, self._b,
                                          assembled=lambda kind: me()._assembled_family(kind))
        self._structure_key = None
        self._eptr = None

    def _get(self, name, make):
        if name not in self._lazy:
            self._lazy[name] = make()
        return self._lazy[name]

    def _assemble(self):
        """Node features and labels of the batch's subjects -- and the block-offset rows and per-node
        normaliser of every model family the cache has built -- in ONE launch (cgnn_gather_rows), on
        first access of any of them.  -> (x, labels, {kind: (blk_off_dst, blk_off_src, norm)})"""
        if "asm" not in self._lazy:
            cache, ds, b = self._cache, self._cache.dataset, self._b
            dev = ds.x.device
            x = torch.empty(b * cache.n, ds.x.shape[2], dtype=ds.x.dtype, device=dev)
            y = torch.empty(b, dtype=ds.labels.dtype, device=dev)
            pairs = [(ds.x, x), (ds.labels, y)]
            fams = {}
            for kind in list(cache._fam):
                fams[kind] = self._family_buffers(kind, pairs)
            self._gather(pairs)
            self._lazy["asm"] = (x, y, fams)
        return self._lazy["asm"]

    def _family_buffers(self, kind, pairs):
        cache, b = self._cache, self._b
        dev = cache.dataset.x.device
        fam = cache.family(kind)
        od = torch.empty(b * (cache.nb + 1), dtype=torch.int32, device=dev)
        os_ = torch.empty(b * (cache.nb + 1), dtype=torch.int32, device=dev)
        norm = torch.empty(b * cache.n, dtype=torch.float32, device=dev)
        pairs += [(fam.blk_off_dst, od), (fam.blk_off_src, os_), (fam.norm, norm)]
        return od, os_, norm

    def _gather(self, pairs):
        dev = self._cache.dataset.x.device
        for lo in range(0, len(pairs), _lib.GATHER_MAX_JOBS):
            jobs = _lib.CgnnGatherJobs()
            chunk = pairs[lo:lo + _lib.GATHER_MAX_JOBS]
            jobs.n = len(chunk)
            for i, (src, dst) in enumerate(chunk):
                assert src.is_contiguous()
                jobs.src[i], jobs.dst[i] = src.data_ptr(), dst.data_ptr()
                jobs.row_bytes[i] = src[0].numel() * src.element_size()
            with _lib.device_guard(dev):
                _lib.check(_lib.load().cgnn_gather_rows(jobs, _lib.ptr(self._ids), self._b, _lib.ptr(self._ids_offset),
                                                        _lib.stream_ptr(dev)), "cgnn_gather_rows")

    def _assembled_family(self, kind):
        """A family's (blk_off_dst, blk_off_src, norm) of this batch; a family the cache builds only now
        (first GraphSAGE batch on a cache that served GCN so far) gets a launch of its own."""
        fams = self._assemble()[2]
        if kind not in fams:
            pairs = []
            fams[kind] = self._family_buffers(kind, pairs)
            self._gather(pairs)
        return fams[kind]

    _ids = property(lambda self: self._get("ids", lambda: self._ids_src.to(device=self._cache.dataset.x.device, dtype=torch.long).contiguous()))
    node_features = property(lambda self: self._assemble()[0])
    labels = property(lambda self: self._assemble()[1])
    ptr = property(lambda self: self._get(
        "ptr", lambda: torch.arange(self._b + 1, device=self._cache.dataset.x.device, dtype=torch.long) * self._cache.n))
    num_graphs = property(lambda self: self._b)
    num_nodes = property(lambda self: self._b * self._cache.n)

    def _materialise(self):
        if self._ids_offset is not None:
            raise RuntimeError("a cursor-addressed ResidentBatch (captured epoch replay) has no fixed COO fields")
        if self._coo is None:
            from .resident import assemble_batch
            full = assemble_batch(self._cache.dataset, self._ids)
            self._coo = (full.edge_index, full.edge_weight, full.batch)
        return self._coo

    edge_index = property(lambda self: self._materialise()[0])
    edge_weight = property(lambda self: self._materialise()[1])
    batch = property(lambda self: self._materialise()[2])

    def structure(self):
        return self._structure

    def invalidate(self) -> None:
        raise RuntimeError("a ResidentBatch's structure belongs to the subject cache")

    def to(self, device) -> "ConnectomeBatch":
        if torch.device(device).type == self._cache.dataset.x.device.type:
            return self
        ei, ew, bt = self._materialise()
        return ConnectomeBatch(self.node_features.to(device), ei.to(device), ew.to(device), bt.to(device),
                               self.labels.to(device), self.ptr.to(device))
