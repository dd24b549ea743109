"""pytest configuration: registers the `gpu` marker and puts the repo root on sys.path.

`-m "not gpu"` runs on a CPU-only box (oracle vs goldens, host logic, C-ABI symbol check);
`-m gpu` runs on an MI355X and calls the HIP kernels through the C ABI.
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a ROCm GPU (MI355X); run with -m gpu")


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no ROCm GPU visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """Always show how many gradient comparisons the fp64 arbiter decided (tests/parity.py)."""
    try:
        from tests import parity as P
    except Exception:      # pragma: no cover
        return
    if P.ARBITER["checked"]:
        terminalreporter.write_line(
            f"[parity] gradient tensors checked: {P.ARBITER['checked']}; decided by the fp64 arbiter: "
            f"{P.ARBITER['fp64']} (of which by the fp32 noise floor: {P.ARBITER['floor']})")
        for line in P.ARBITER["names"]:
            terminalreporter.write_line(f"   arbiter: {line}")
        for line in P.ARBITER.get("relu_ties", []):
            terminalreporter.write_line(f"   relu tie: {line}")
