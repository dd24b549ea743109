"""Synthetic Watts-Strogatz connectomes (data preparation; not on the timed path).

Same public API and the same distributions as the reference generator
(connectome_gnn/synthetic.py:97-301): ring lattice of degree k, each lattice edge rewired with
probability beta to a uniformly chosen non-neighbour, one Beta(2,5) weight per undirected
edge stored in both directions, five node features, a noisy linear binary label.  It is NOT
stream-identical to the reference (that one iterates a Python set of tuples, whose order is
an implementation detail, and is O(E^2) per graph -- 87 s for one 1000-ROI graph); golden
G8 keeps one reference-generated graph as data instead.

``generate_packed`` builds a whole regular dataset as dense arrays (every WS graph has exactly
n*k directed edges) for the device-resident loader in ``resident.py``.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional

import numpy as np
import torch

from .graph import ConnectomeGraph

# Abbreviated Desikan-Killiany style parcellation: 30 cortical areas per hemisphere plus
# subcortical structures and summary tracts (FreeSurfer naming).  The reference's list
# (synthetic.py:38-88) is documented as "84 ROIs" but holds 83 entries, so its default
# num_regions is 83; the same 83 names in the same order are kept here.
_CORTICAL = [
    "superiorfrontal", "rostralmiddlefrontal", "caudalmiddlefrontal", "parsopercularis",
    "parsorbitalis", "parstriangularis", "lateralorbitofrontal", "medialorbitofrontal",
    "precentral", "superiorparietal", "inferiorparietal", "supramarginal", "postcentral",
    "precuneus", "posteriorcingulate", "isthmuscingulate", "superiortemporal", "middletemporal",
    "inferiortemporal", "fusiform", "entorhinal", "parahippocampal", "transversetemporal",
    "lateraloccipital", "lingual", "cuneus", "pericalcarine", "rostralanteriorcingulate",
    "caudalanteriorcingulate", "paracingulate",
]
_SUBCORTICAL = ["Thalamus", "Caudate", "Putamen", "Pallidum", "Hippocampus", "Amygdala",
                "Accumbens-area"]
REGION_NAMES: List[str] = (
    [f"ctx-{h}-{a}" for a in _CORTICAL for h in ("lh", "rh")]
    + [f"{side}-{s}" for s in _SUBCORTICAL for side in ("Left", "Right")]
    + ["Brain-Stem", "CC_anterior", "CC_posterior", "UncF_left", "UncF_right", "ILF_left",
       "ILF_right", "CST_left", "CST_right"]
)
NUM_REGIONS = len(REGION_NAMES)


def _ws_pairs(n: int, k: int, beta: float, rng: np.random.Generator):
    """Undirected Watts-Strogatz pairs (u[i], v[i]), i < n*(k//2); edge count is preserved."""
    half = k // 2
    u = np.repeat(np.arange(n, dtype=np.int64), half)
    v = (u + np.tile(np.arange(1, half + 1, dtype=np.int64), n)) % n
    m = u.shape[0]
    if beta <= 0.0 or m == 0:
        return u, v
    adj = bytearray(n * n)
    ul, vl = u.tolist(), v.tolist()
    deg = [0] * n
    for a, b in zip(ul, vl):
        if not adj[a * n + b]:
            adj[a * n + b] = adj[b * n + a] = 1
            deg[a] += 1
            deg[b] += 1
    picks = np.flatnonzero(rng.random(m) < beta).tolist()
    draws = rng.integers(0, n, size=4 * len(picks) + 16).tolist()
    di = 0
    for i in picks:
        a, b = ul[i], vl[i]
        if deg[a] >= n - 1:          # no free target: keep the lattice edge
            continue
        adj[a * n + b] = adj[b * n + a] = 0
        deg[a] -= 1
        deg[b] -= 1
        while True:                  # uniform over non-neighbours of a by rejection
            if di == len(draws):
                draws = rng.integers(0, n, size=1024).tolist()
                di = 0
            w = draws[di]
            di += 1
            if w != a and not adj[a * n + w]:
                break
        adj[a * n + w] = adj[w * n + a] = 1
        deg[a] += 1
        deg[w] += 1
        vl[i] = w
    return u, np.asarray(vl, dtype=np.int64)


def _graph_arrays(n: int, k: int, beta: float, trait_idx: int, rng: np.random.Generator):
    u, v = _ws_pairs(n, k, beta, rng)
    m = u.shape[0]
    wt = rng.beta(2.0, 5.0, size=m).astype(np.float32)
    src = np.empty(2 * m, dtype=np.int64)
    dst = np.empty(2 * m, dtype=np.int64)
    src[0::2], src[1::2] = u, v          # both directions stored next to each other
    dst[0::2], dst[1::2] = v, u
    w = np.repeat(wt, 2)
    # node features (reference synthetic.py:150-183)
    deg = np.bincount(src, weights=w, minlength=n).astype(np.float32)
    cnt = np.bincount(src, minlength=n).astype(np.float32)
    deg_norm = deg / (deg.max() + 1e-8)
    cluster = deg / (cnt + 1e-8)
    vol = rng.lognormal(7.5, 0.5, size=n).astype(np.float32)
    vol = (vol - vol.mean()) / (vol.std(ddof=1) + 1e-8)
    act = rng.normal(0.0, 1.0, size=n).astype(np.float32)
    thick = np.clip(rng.normal(2.5, 0.3, size=n), 1.5, 4.0).astype(np.float32)
    thick = (thick - thick.mean()) / (thick.std(ddof=1) + 1e-8)
    x = np.stack([deg_norm, cluster, vol, act, thick], axis=1).astype(np.float32)
    # label (reference synthetic.py:194-215)
    tw = np.random.default_rng(trait_idx * 1337).normal(0.0, 1.0, 3)
    score = tw[0] * float(deg_norm.mean()) + tw[1] * float(w.mean()) + tw[2] * float(cluster.mean())
    score += rng.normal(0.0, 2.0)
    return x, np.stack([src, dst]), w, int(score > 0)


def generate_connectome(num_regions: int = NUM_REGIONS, k: int = 8, beta: float = 0.15,
                        trait_idx: int = 0, subject_id: Optional[str] = None,
                        seed: Optional[int] = None) -> ConnectomeGraph:
    """One synthetic subject (reference synthetic.py:222-263)."""
    rng = np.random.default_rng(seed)
    if subject_id is None:
        subject_id = f"sub-{rng.integers(10000, 99999)}"
    x, ei, w, label = _graph_arrays(num_regions, k, beta, trait_idx, rng)
    return ConnectomeGraph(torch.from_numpy(x), torch.from_numpy(ei), torch.from_numpy(w),
                           torch.tensor(label, dtype=torch.long), subject_id)


def generate_dataset(num_subjects: int = 200, num_regions: int = NUM_REGIONS, k: int = 8,
                     beta: float = 0.15, trait_idx: int = 0, seed: int = 42) -> list:
    """``num_subjects`` subjects; per-subject seeds drawn from the master seed
    (reference synthetic.py:266-301)."""
    seeds = np.random.default_rng(seed).integers(0, 2 ** 31, size=num_subjects).tolist()
    return [generate_connectome(num_regions, k, beta, trait_idx, f"sub-{i:04d}", int(seeds[i]))
            for i in range(num_subjects)]


@dataclass
class PackedDataset:
    """A regular dataset as dense arrays: S subjects, n nodes and e directed edges each."""
    x: torch.Tensor             # [S, n, F]  f32
    edge_local: torch.Tensor    # [S, 2, e]  i64, node ids local to the graph
    edge_weight: torch.Tensor   # [S, e]     f32
    labels: torch.Tensor        # [S]        i64

    @property
    def num_subjects(self) -> int:
        return int(self.x.shape[0])

    def to(self, device) -> "PackedDataset":
        return PackedDataset(self.x.to(device), self.edge_local.to(device),
                             self.edge_weight.to(device), self.labels.to(device))

    def graph(self, i: int) -> ConnectomeGraph:
        return ConnectomeGraph(self.x[i], self.edge_local[i], self.edge_weight[i], self.labels[i],
                               f"sub-{i:04d}")

    def relabel_by_degree(self) -> "PackedDataset":
        """The same graphs with every subject's nodes renumbered by decreasing degree (in + out,
        ties in the old order).  A GCN / GraphSAGE with a mean-pool readout is invariant under it
        (same logits up to the order of floating-point sums); what changes is the padding of the
        blocked-ELL the fused kernels walk, which pads every row to the widest of its 16-row block:
        19 % of the steps on 360-ROI small-world graphs in node order, 3 % in degree order."""
        S, n, f = self.x.shape
        dev = self.x.device
        src, dst = self.edge_local[:, 0], self.edge_local[:, 1]
        one = torch.ones_like(src)
        deg = torch.zeros(S, n, dtype=torch.long, device=dev).scatter_add_(1, dst, one).scatter_add_(1, src, one)
        perm = torch.argsort(deg, dim=1, descending=True, stable=True)          # new id -> old id
        inv = torch.empty_like(perm).scatter_(1, perm, torch.arange(n, device=dev).expand(S, n))
        x = torch.gather(self.x, 1, perm[..., None].expand(S, n, f))
        e = torch.stack([torch.gather(inv, 1, src), torch.gather(inv, 1, dst)], 1)
        return PackedDataset(x.contiguous(), e.contiguous(), self.edge_weight, self.labels)

    @staticmethod
    def from_graphs(graphs) -> "PackedDataset":
        """Pack a list of ConnectomeGraphs (e.g. the reference's ``generate_dataset`` output) that
        all have the same number of nodes and edges and carry labels: the dataset can then live
        in HBM and be batched by ``resident.ResidentDataLoader`` without host work."""
        if not graphs:
            raise ValueError("empty dataset")
        n, e = graphs[0].num_nodes, graphs[0].num_edges
        for g in graphs:
            if g.num_nodes != n or g.num_edges != e:
                raise ValueError("PackedDataset needs graphs of one size (nodes and edges)")
            if g.label is None:
                raise ValueError("PackedDataset needs labelled graphs")
        return PackedDataset(torch.stack([g.node_features for g in graphs]),
                             torch.stack([g.edge_index for g in graphs]),
                             torch.stack([g.edge_weight for g in graphs]),
                             torch.stack([torch.as_tensor(g.label, dtype=torch.long).reshape(()) for g in graphs]))


def _packed_arrays(args):
    seeds, num_regions, k, beta, trait_idx = args
    e = num_regions * (k // 2) * 2
    xs = np.empty((len(seeds), num_regions, 5), dtype=np.float32)
    eis = np.empty((len(seeds), 2, e), dtype=np.int64)
    ws = np.empty((len(seeds), e), dtype=np.float32)
    ys = np.empty(len(seeds), dtype=np.int64)
    for i, sd in enumerate(seeds):
        rng = np.random.default_rng(int(sd))
        xs[i], eis[i], ws[i], ys[i] = _graph_arrays(num_regions, k, beta, trait_idx, rng)
    return xs, eis, ws, ys


def generate_packed(num_subjects: int, num_regions: int = NUM_REGIONS, k: int = 8,
                    beta: float = 0.15, trait_idx: int = 0, seed: int = 42, workers: int = 1) -> PackedDataset:
    """Same graphs as ``generate_dataset`` (same seeds), packed into dense arrays.  workers > 1: the
    subjects (independent, one seed each) are generated by that many forked processes -- the same arrays;
    call it before the process touches the GPU (BASELINE config 4's 32,768 x 360-ROI dataset takes 100 s on
    one core)."""
    seeds = np.random.default_rng(seed).integers(0, 2 ** 31, size=num_subjects).tolist()
    if workers <= 1 or num_subjects < 4 * workers:
        parts = [_packed_arrays((seeds, num_regions, k, beta, trait_idx))]
    else:
        import multiprocessing as mp
        step = -(-num_subjects // (4 * workers))
        jobs = [(seeds[lo:lo + step], num_regions, k, beta, trait_idx) for lo in range(0, num_subjects, step)]
        with mp.get_context("fork").Pool(workers) as pool:
            parts = pool.map(_packed_arrays, jobs)
    xs, eis, ws, ys = (np.concatenate([p[j] for p in parts]) for j in range(4))
    return PackedDataset(torch.from_numpy(xs), torch.from_numpy(eis), torch.from_numpy(ws),
                         torch.from_numpy(ys))


def small_world_stats(graphs: list) -> dict:
    """Mean clustering coefficient and characteristic path length (reference
    synthetic.py:304-339): weighted triangle fraction per node; BFS from <= 20 start nodes."""
    clus, paths = [], []
    for g in graphs:
        a = g.adjacency_matrix().numpy().astype(np.float64)
        n = a.shape[0]
        deg = a.sum(1)
        tri = np.einsum("ij,jk,ki->i", a, a, a)
        denom = deg * (deg - 1.0)
        c = np.divide(tri, denom, out=np.zeros(n), where=denom > 0)
        clus.append(float(c.mean()))
        nbrs = [np.flatnonzero(a[i] > 0) for i in range(n)]
        dists = []
        for s in range(min(20, n)):
            dist = np.full(n, -1, dtype=np.int64)
            dist[s] = 0
            frontier = [s]
            while frontier:
                nxt = []
                for u in frontier:
                    for v in nbrs[u]:
                        if dist[v] < 0:
                            dist[v] = dist[u] + 1
                            nxt.append(int(v))
                frontier = nxt
            dists.extend(dist[dist > 0].tolist())
        paths.append(float(np.mean(dists)) if dists else float("nan"))
    return {"mean_clustering": float(np.mean(clus)),
            "mean_avg_path_length": float(np.nanmean(paths)), "num_graphs": len(graphs)}
