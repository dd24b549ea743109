#!/usr/bin/env python3
"""Diagnostic only: where does a tile's time go inside the fused GCN kernels?

Builds nothing itself: run with CGNN_LIB pointing at a -DCGNN_STAMPS build
(make -C connectome_gnn_amd/csrc stamps).  Prints, per kernel, the share of wave-cycles spent in
each stamped phase (s_memtime deltas accumulated per wave).  Read SHARES, not lengths: the
stamps' own waits forbid overlaps the shipped kernels have (cdna_hip_programming.md section 7).
"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import connectome_gnn_amd as C  # noqa: E402
from connectome_gnn_amd import _lib  # noqa: E402
from connectome_gnn_amd.resident import assemble_batch  # noqa: E402
from connectome_gnn_amd.synthetic import generate_packed  # noqa: E402

NAMES = ["A:load+fill", "A:barrier", "B:meta commit", "B:aggregate", "B:stage+Afrag (bwd: mfma+epi)",
         "B:mfma (fwd)", "end barrier", "B:epilogue (fwd)"]


def read(lib):
    buf = np.zeros(1024 * 8 * 8, dtype=np.uint64)
    rc = lib.cgnn_debug_stamps(buf.ctypes.data_as(ctypes.c_void_p))
    assert rc == 0
    return buf.reshape(1024, 8, 8)[:256]


def show(tag, st):
    tot = st.sum()
    per = st.sum(axis=(0, 1)) / max(tot, 1)
    cyc = st.sum(axis=2)                      # [wg][wave] total cycles
    print(f"{tag}: mean wave-cycles/WG-wave {cyc.mean():.0f}  (100 MHz ticks x clock ratio)")
    for n, p in zip(NAMES, per):
        if p > 0:
            print(f"    {n:18s} {100 * p:5.1f} %")


def main():
    lib = _lib.load()
    lib.cgnn_debug_stamps.restype = ctypes.c_int
    lib.cgnn_debug_stamps.argtypes = [ctypes.c_void_p]
    nb = int(os.environ.get("PROBE_GRAPHS", "4096"))
    ds = generate_packed(min(nb, 512), 360, 14, seed=1).to("cuda")
    b = assemble_batch(ds, torch.arange(nb) % min(nb, 512))
    torch.manual_seed(0)
    m = C.GCNConnectome(5, 64, dropout=float(os.environ.get("PROBE_DROPOUT", "0.3")),
                        impl="fused").to("cuda").train()
    for _ in range(2):
        m(b).sum().backward()
    read(lib)
    # forward only: layer 0 kernel + two generic kernels
    out = m(b)
    st_f = read(lib)
    show("forward (fwd_first + 2x fwd)", st_f)
    if os.environ.get("PROBE_NODROP"):
        return
    out.sum().backward()
    st_b = read(lib)
    show("backward (2x bwd + bwd_first)", st_b)


if __name__ == "__main__":
    main()
