"""Device-side structure of a ConnectomeBatch: what the HIP kernels index with.

The reference never builds this -- it re-derives everything from the int64 COO inside every
layer call (models.py:94-113, 146-149).  Here the COO is bucket-sorted once per batch into a
destination-sorted CSR (forward segment sums) and a source-sorted CSR (their transposes in
backward), both int32 and stable (COO order inside a row), by ``cgnn_csr_build``.

All arrays are torch tensors only so that their memory comes from torch's caching
allocator; every number in them is produced by the HIP library.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch

from . import _lib


def _require_device(t: torch.Tensor, what: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(
            f"{what} is on {t.device}: connectome_gnn_amd runs the message-passing path on a "
            "ROCm device only (there is no CPU fallback; move the batch/model with .to('cuda')).")


@dataclass
class GcnNorm:
    dis: torch.Tensor        # [Nn]  (deg + 1e-8)^-1/2, source-side degree incl. self-loop
    selfc: torch.Tensor      # [Nn]  dis^2  (coefficient of the appended self-loop)
    coef_dst: torch.Tensor   # [Ee]  c_e in dst-CSR slot order
    coef_src: torch.Tensor   # [Ee]  c_e in src-CSR slot order


@dataclass
class SageNorm:
    den: torch.Tensor            # [Nn]  sum of in-edge weights + 1e-8
    w_dst: torch.Tensor          # [Ee]  w_e in dst-CSR slot order
    coef_src_bwd: torch.Tensor   # [Ee]  w_e / den[dst_e] in src-CSR slot order


TILED_MAX_ROWS = 384      # CGNN_FUSED_MAX_ROWS: rows of one LDS tile


def twin_view(structure, x: torch.Tensor):
    """(structure', x', twin): the batch's degree-ordered twin and the node features in its order when
    one has been prepared (``BatchStructure.degree_ordered_twin`` via ``model.prepare_batch(batch,
    reuse=True)``), else the inputs and None.  The one-node encoders call this on entry: all their
    per-node arrays are internal, so they run on the twin unchanged."""
    twin = getattr(structure, "__dict__", {}).get("_degree_twin")
    if twin is None:
        return structure, x, None
    return twin, twin.permuted_features(x), twin


def call_prepare(prepare, batch) -> None:
    """``prepare(batch, reuse=True)`` when the callable takes ``reuse`` (connectome_gnn_amd models'
    ``prepare_batch``), else ``prepare(batch)`` -- decided from the signature, so a TypeError raised INSIDE
    the callable is never mistaken for a missing keyword (and the callable never runs twice)."""
    import inspect
    try:
        params = inspect.signature(prepare).parameters
        takes = "reuse" in params or any(p.kind is inspect.Parameter.VAR_KEYWORD for p in params.values())
    except (TypeError, ValueError):          # builtins / C callables without a signature
        takes = False
    if takes:
        prepare(batch, reuse=True)
    else:
        prepare(batch)


def unpermute_record(twin, rec) -> None:
    """Keep bytes recorded per node of the twin -> the batch's node order (parity hook)."""
    if twin is None or rec is None or rec.get("layers") is None:
        return
    inv = torch.empty_like(twin.perm)
    inv[twin.perm] = torch.arange(twin.perm.numel(), device=twin.perm.device)
    rec["layers"] = [None if m is None else m.view(twin.num_nodes, -1).index_select(0, inv).reshape(-1)
                     for m in rec["layers"]]


@dataclass
class FusedMeta:
    """Static per-batch metadata of the fused per-tile kernels (see include/cgnn.h)."""
    tile_ptr: torch.Tensor      # int32 [T+1]
    tile_blk: torch.Tensor      # int32 [T+1] first 16-row block of each tile
    max_tile_rows: int
    num_blocks: int
    blk_off_dst: torch.Tensor   # int32 [NB+1] (entries)
    ent_dst: torch.Tensor       # uint8 [8 * entries]  blocked-ELL, rows = destinations
    blk_off_src: torch.Tensor
    ent_src: torch.Tensor       # blocked-ELL, rows = sources
    w_src: torch.Tensor         # f32 [Ee] edge weights in src-CSR slot order (for dis)


class BatchStructure:
    """dst-/src-sorted CSR (+ per-graph node ranges) of one batch, on its device."""

    def __init__(self):
        self.num_nodes = 0
        self.num_edges = 0
        self.num_graphs = 0
        self.max_nodes_per_graph = 0
        self.max_in_degree = 0
        self.max_out_degree = 0
        self.block_diagonal = True
        self.gptr: Optional[torch.Tensor] = None         # int32 [B+1]
        self.node_graph: Optional[torch.Tensor] = None   # int32 [Nn]
        self.rowptr_dst = self.eid_dst = self.col_dst = None
        self.rowptr_src = self.eid_src = self.col_src = None
        self._edge_index = None
        self._edge_weight = None
        self._ptr_host = None
        self._tiles = {}

    @staticmethod
    def build(batch, force_generic: bool = False) -> "BatchStructure":
        ei, ew = batch.edge_index, batch.edge_weight
        _require_device(ei, "batch.edge_index")
        if ei.dtype != torch.int64 or ei.dim() != 2 or ei.shape[0] != 2:
            raise ValueError("edge_index must be int64 [2, E]")
        if ew.dtype != torch.float32:
            raise TypeError(f"edge_weight must be float32 (the reference is fp32-only), got {ew.dtype}")
        lib = _lib.load()
        dev = ei.device
        ei = ei.contiguous()
        nn_, ne = batch.num_nodes, int(ei.shape[1])
        s = BatchStructure()
        s.num_nodes, s.num_edges, s.num_graphs = nn_, ne, batch.num_graphs
        s._edge_index, s._edge_weight = ei, ew.contiguous()
        i32 = dict(dtype=torch.int32, device=dev)
        s.rowptr_dst = torch.empty(nn_ + 1, **i32)
        s.rowptr_src = torch.empty(nn_ + 1, **i32)
        s.eid_dst, s.col_dst = torch.empty(ne, **i32), torch.empty(ne, **i32)
        s.eid_src, s.col_src = torch.empty(ne, **i32), torch.empty(ne, **i32)
        flags = torch.empty(4, **i32)
        node_graph = batch.batch.contiguous() if batch.batch is not None else None
        if node_graph is not None and (node_graph.dtype != torch.int64 or node_graph.numel() != nn_):
            raise ValueError("batch.batch must be int64 [num_nodes]")
        s._ptr_host = batch.ptr.detach().cpu().numpy().astype(np.int64)
        sizes = np.diff(s._ptr_host)
        s.max_nodes_per_graph = int(sizes.max()) if sizes.size else 0
        with _lib.device_guard(dev):
            s.gptr = batch.ptr.to(device=dev, dtype=torch.int32)
            s.node_graph = (node_graph.to(torch.int32) if node_graph is not None
                            else torch.zeros(nn_, dtype=torch.int32, device=dev))
            f = None
            eptr = getattr(batch, "_eptr", None)
            s.__dict__["_eptr_host"] = eptr
            if (not force_generic and eptr is not None and eptr.numel() == batch.num_graphs + 1
                    and int(eptr[-1]) == ne and batch.num_graphs > 0):
                # COO grouped by graph: whole graphs are built in LDS by one workgroup each
                emax = int((eptr[1:] - eptr[:-1]).max())
                rc = lib.cgnn_csr_build_grouped(
                    _lib.ptr(ei), _lib.ptr(s.gptr), _lib.ptr(eptr.to(device=dev, dtype=torch.int32)),
                    batch.num_graphs, nn_, ne, s.max_nodes_per_graph, emax,
                    _lib.ptr(s.rowptr_dst), _lib.ptr(s.eid_dst), _lib.ptr(s.col_dst),
                    _lib.ptr(s.rowptr_src), _lib.ptr(s.eid_src), _lib.ptr(s.col_src),
                    _lib.ptr(flags), _lib.stream_ptr(dev))
                if rc == _lib.CGNN_OK:
                    f = flags.tolist()              # one sync per batch, at build time only
                    if f[0] or f[1]:
                        f = None                    # not grouped after all: generic build decides
                elif rc != _lib.CGNN_EUNSUPPORTED:
                    _lib.check(rc, "cgnn_csr_build_grouped")
            if f is None:
                ws = torch.empty(int(lib.cgnn_csr_workspace_bytes(nn_, ne)), dtype=torch.uint8, device=dev)
                _lib.check(lib.cgnn_csr_build(
                    _lib.ptr(ei), _lib.ptr(node_graph), nn_, ne,
                    _lib.ptr(s.rowptr_dst), _lib.ptr(s.eid_dst), _lib.ptr(s.col_dst),
                    _lib.ptr(s.rowptr_src), _lib.ptr(s.eid_src), _lib.ptr(s.col_src),
                    _lib.ptr(flags), _lib.ptr(ws), _lib.nbytes(ws), _lib.stream_ptr(dev)), "cgnn_csr_build")
                f = flags.tolist()                  # one sync per batch, at build time only
        if f[0]:
            # the reference would raise from scatter_add_/index (models.py:104,112)
            raise IndexError(f"{f[0]} edge(s) reference a node outside [0, {nn_})")
        s.block_diagonal = f[1] == 0
        s.max_in_degree, s.max_out_degree = f[2], f[3]
        return s

    # -- internal node order of the fused per-tile GCN path ---------------------------------
    def degree_ordered_twin(self) -> "BatchStructure":
        """The same batch with every graph's nodes renumbered by decreasing degree (in + out, ties in
        the old order), as a structure of its own; ``twin.perm`` (int64 [Nn]) maps its node ids to
        this structure's.  The blocked-ELL pads every row to the widest row of its 16-row block:
        19 % of the steps the tile kernels walk on 360-ROI small-world graphs in node order, 3 % in
        degree order.  A GCN with a mean-pool readout is invariant under the renumbering, and every
        per-node array of the fused encoder is internal to it, so the encoder can run on the twin
        (node features gathered through ``perm`` on entry) while the batch's public arrays stay as
        they are.  Built once per batch on request (`model.prepare_batch(batch, reuse=True)`): it
        costs a second CSR / blocked-ELL build, which pays for batches that are trained on repeatedly."""
        twin = self.__dict__.get("_degree_twin")
        if twin is not None:
            return twin
        ei = self._edge_index
        dev = ei.device
        nn_ = self.num_nodes
        one = torch.ones(ei.shape[1], dtype=torch.long, device=dev)
        deg = torch.zeros(nn_, dtype=torch.long, device=dev).index_add_(0, ei[0], one).index_add_(0, ei[1], one)
        gid = self.node_graph.to(torch.long)
        # one stable sort by (graph asc, degree desc): nodes stay inside their graph's run
        key = gid * (int(deg.max()) + 1 if nn_ else 1) + (deg.max() - deg if nn_ else deg)
        perm = torch.argsort(key, stable=True)                     # twin id -> this id
        inv = torch.empty_like(perm)
        inv[perm] = torch.arange(nn_, device=dev)

        class _Shim:                                               # what BatchStructure.build reads of a batch
            pass
        shim = _Shim()
        shim.edge_index = inv[ei]                                  # same COO order: still grouped by graph
        shim.edge_weight = self._edge_weight
        shim.num_nodes, shim.num_graphs = nn_, self.num_graphs
        shim.batch = gid
        shim.ptr = self.gptr.to(torch.long)
        shim._eptr = self.__dict__.get("_eptr_host")
        twin = BatchStructure.build(shim)
        twin.perm = perm
        twin.__dict__["_is_twin"] = True
        self.__dict__["_degree_twin"] = twin
        return twin

    def permuted_features(self, x: torch.Tensor) -> torch.Tensor:
        """Node features in this (twin) structure's node order; cached per feature tensor version."""
        key = (x.data_ptr(), x._version, tuple(x.shape))
        hit = self.__dict__.get("_xperm")
        if hit is None or hit[0] != key:
            hit = (key, x.index_select(0, self.perm).contiguous())
            self.__dict__["_xperm"] = hit
        return hit[1]

    # -- dense fragments of the operators of large graphs (band_aggregate.hip) -----------------
    def band_ops(self, kind: str, norm):
        """(fwd, bwd) `(self, ops.BandOp)` pairs -- or (None, None) -- of the GCN / GraphSAGE operator:
        built once per batch from the first normalisation computed for it (the coefficients are a
        function of the batch's edge weights only), like the CSR and the blocked-ELL."""
        hit = self.__dict__.setdefault("_band_ops", {})
        if kind not in hit:
            from . import ops
            if kind == "gcn":
                f = ops.band_operator_f32(self, self.rowptr_dst, self.col_dst, norm.coef_dst)
                b = ops.band_operator_f32(self, self.rowptr_src, self.col_src, norm.coef_src) if f is not None else None
            else:
                f = ops.band_operator_f32(self, self.rowptr_dst, self.col_dst, norm.w_dst)
                b = ops.band_operator_f32(self, self.rowptr_src, self.col_src, norm.coef_src_bwd) if f is not None else None
            hit[kind] = (f, b) if (f is not None and b is not None) else None
        # (the pairs are made per call: a cached tuple holding `self` would be a reference cycle, and a
        # structure with its device arrays would then wait for the cycle collector instead of its refcount)
        return (None, None) if hit[kind] is None else ((self, hit[kind][0]), (self, hit[kind][1]))

    # -- normalisations: layer independent, recomputed once per forward pass ----------------
    def gcn_norm(self) -> GcnNorm:
        """models.py:94-108 via cgnn_gcn_norm."""
        lib = _lib.load()
        dev = self.rowptr_dst.device
        f32 = dict(dtype=torch.float32, device=dev)
        n = GcnNorm(torch.empty(self.num_nodes, **f32), torch.empty(self.num_nodes, **f32),
                    torch.empty(self.num_edges, **f32), torch.empty(self.num_edges, **f32))
        with _lib.device_guard(dev):
            _lib.check(lib.cgnn_gcn_norm(
                _lib.ptr(self._edge_index), _lib.ptr(self._edge_weight), self.num_nodes,
                self.num_edges, _lib.ptr(self.rowptr_dst), _lib.ptr(self.eid_dst),
                _lib.ptr(self.rowptr_src), _lib.ptr(self.eid_src), _lib.ptr(n.dis),
                _lib.ptr(n.selfc), _lib.ptr(n.coef_dst), _lib.ptr(n.coef_src),
                _lib.stream_ptr()), "cgnn_gcn_norm")
        return n

    def sage_norm(self, backward_coef: bool = True) -> SageNorm:
        """models.py:146-149 via cgnn_sage_norm.  backward_coef=False skips the per-edge
        w_e / den[dst_e] array of the gather-form backward (the tiled form divides by den itself)."""
        lib = _lib.load()
        dev = self.rowptr_dst.device
        f32 = dict(dtype=torch.float32, device=dev)
        n = SageNorm(torch.empty(self.num_nodes, **f32), torch.empty(self.num_edges, **f32),
                     torch.empty(self.num_edges, **f32) if backward_coef else None)
        with _lib.device_guard(dev):
            _lib.check(lib.cgnn_sage_norm(
                _lib.ptr(self._edge_index), _lib.ptr(self._edge_weight), self.num_nodes,
                self.num_edges, _lib.ptr(self.rowptr_dst), _lib.ptr(self.eid_dst),
                _lib.ptr(self.rowptr_src), _lib.ptr(self.eid_src), _lib.ptr(n.den),
                _lib.ptr(n.w_dst), _lib.ptr(n.coef_src_bwd), _lib.stream_ptr()), "cgnn_sage_norm")
        return n

    # -- tiling for the fused per-tile kernels ----------------------------------------------
    def tile_ptr(self, max_rows: int, num_workgroups: int) -> torch.Tensor:
        """int32 [T+1] node offsets of tiles = runs of consecutive whole graphs with at most
        ``max_rows`` nodes (LDS capacity).  The row cap is lowered when the batch is small so
        that every persistent workgroup gets a tile."""
        key = (max_rows, num_workgroups)
        if key not in self._tiles:
            ptr = self._ptr_host
            if self.max_nodes_per_graph > max_rows:
                raise ValueError("a graph exceeds the tile capacity")
            # small batches: about one tile per persistent workgroup (measured best at 512 x 84-ROI:
            # fewer, fuller tiles beat more balance slack when the step is latency-bound)
            cap = min(max_rows, max(self.max_nodes_per_graph,
                                    -(-self.num_nodes // max(num_workgroups, 1))))
            sizes = np.diff(ptr)
            if sizes.size and (sizes == sizes[0]).all():
                per = max(1, cap // max(int(sizes[0]), 1))
                cuts = ptr[::per]
                if cuts[-1] != ptr[-1]:
                    cuts = np.append(cuts, ptr[-1])
            else:
                # greedy: a tile takes as many whole graphs as fit under the cap -- one binary
                # search per TILE over the node offsets (not a Python step per graph)
                cuts, start, end, g = [0], 0, int(ptr[-1]), 0
                while start < end:
                    g = int(np.searchsorted(ptr, start + cap, side="right")) - 1   # ptr[g] <= start + cap
                    if int(ptr[g]) <= start:
                        raise ValueError("a graph exceeds the tile capacity")
                    start = int(ptr[g])
                    cuts.append(start)
                cuts = np.asarray(cuts, dtype=np.int64)
            rows = int(np.diff(cuts).max()) if cuts.size > 1 else 0
            t = torch.from_numpy(cuts.astype(np.int32)).to(self.rowptr_dst.device)
            self._tiles[key] = (t, rows)
        return self._tiles[key]

    def fused_meta(self, max_rows: int, num_workgroups: int, self_weight: float = 1.0) -> FusedMeta:
        """Blocked-ELL metadata for the fused / tiled kernels, built once per batch by the HIP
        library (cgnn_bell_plan / cgnn_bell_fill / cgnn_gather_f32).  self_weight: weight of the
        appended self-loop entry (GCN 1, GraphSAGE 0 = no self-loop)."""
        key = ("meta", max_rows, num_workgroups, float(self_weight))
        if key in self._tiles:
            return self._tiles[key]
        lib = _lib.load()
        dev = self.rowptr_dst.device
        tptr, rows = self.tile_ptr(max_rows, num_workgroups)
        cuts = tptr.cpu().numpy().astype(np.int64)
        nblk = (np.diff(cuts) + 15) // 16
        tile_blk_h = np.concatenate([[0], np.cumsum(nblk)]).astype(np.int32)
        nb = int(tile_blk_h[-1])
        nt = int(cuts.size) - 1
        tile_blk = torch.from_numpy(tile_blk_h).to(dev)
        i32 = dict(dtype=torch.int32, device=dev)
        out = {}
        with _lib.device_guard(dev):
            for name, rowptr, col, eid in (("dst", self.rowptr_dst, self.col_dst, self.eid_dst),
                                           ("src", self.rowptr_src, self.col_src, self.eid_src)):
                blk_off = torch.empty(nb + 1, **i32)
                scratch = torch.empty(nb // 2048 + 8, **i32)
                _lib.check(lib.cgnn_bell_plan(_lib.ptr(tptr), _lib.ptr(tile_blk), nt, nb,
                                              _lib.ptr(rowptr), _lib.ptr(blk_off), _lib.ptr(scratch), _lib.nbytes(scratch),
                                              _lib.stream_ptr()), "cgnn_bell_plan")
                # entries <= 16 * (max degree of the ordering + 1) per block: sized from the degree
                # bound recorded at CSR build, so no read-back (and no stall of the stream) here
                maxdeg = self.max_in_degree if name == "dst" else self.max_out_degree
                total = nb * 16 * (int(maxdeg) + 1)
                if total > 2 ** 31 - 17:
                    raise ValueError("blocked-ELL of this batch exceeds 2^31 entries; use smaller batches")
                ent = torch.empty(max(total, 1) * 8, dtype=torch.uint8, device=dev)
                _lib.check(lib.cgnn_bell_fill(_lib.ptr(tptr), _lib.ptr(tile_blk), nt,
                                              _lib.ptr(rowptr), _lib.ptr(col), _lib.ptr(eid),
                                              _lib.ptr(self._edge_weight), float(self_weight),
                                              _lib.ptr(blk_off), _lib.ptr(ent), _lib.stream_ptr()),
                           "cgnn_bell_fill")
                out[name] = (blk_off, ent)
            w_src = torch.empty(self.num_edges, dtype=torch.float32, device=dev)
            _lib.check(lib.cgnn_gather_f32(_lib.ptr(self._edge_weight), _lib.ptr(self.eid_src),
                                           self.num_edges, _lib.ptr(w_src), _lib.stream_ptr()),
                       "cgnn_gather_f32")
        m = FusedMeta(tptr, tile_blk, rows, nb, out["dst"][0], out["dst"][1], out["src"][0],
                      out["src"][1], w_src)
        self._tiles[key] = m
        return m

    def tiles_struct(self, meta: FusedMeta, dis: Optional[torch.Tensor] = None):
        """Host block of device pointers (`struct cgnn_tiles`, include/cgnn.h) for `meta`."""
        t = _lib.CgnnTiles()
        t.num_nodes = self.num_nodes
        t.num_tiles = int(meta.tile_ptr.numel()) - 1
        t.max_tile_rows = meta.max_tile_rows
        t.tile_ptr, t.tile_blk = meta.tile_ptr.data_ptr(), meta.tile_blk.data_ptr()
        t.blk_off_dst, t.ent_dst = meta.blk_off_dst.data_ptr(), meta.ent_dst.data_ptr()
        t.blk_off_src, t.ent_src = meta.blk_off_src.data_ptr(), meta.ent_src.data_ptr()
        t.dis = dis.data_ptr() if dis is not None else None
        return t

    def tiled_ok(self, width: int) -> bool:
        """The LDS-tiled aggregate (cgnn_aggregate_tiled_f32) covers this batch at this width."""
        return (width % 64 == 0 and self.block_diagonal and self.num_nodes > 0
                and self.max_nodes_per_graph <= TILED_MAX_ROWS)

    def gcn_dis(self, meta: FusedMeta) -> torch.Tensor:
        """dis = (source-side degree + self-loop + 1e-8)^-1/2, models.py:97-105; every step."""
        lib = _lib.load()
        dev = self.rowptr_dst.device
        dis = torch.empty(self.num_nodes, dtype=torch.float32, device=dev)
        with _lib.device_guard(dev):
            _lib.check(lib.cgnn_gcn_dis(_lib.ptr(meta.w_src), _lib.ptr(self.rowptr_src),
                                        self.num_nodes, _lib.ptr(dis), _lib.stream_ptr()),
                       "cgnn_gcn_dis")
        return dis
