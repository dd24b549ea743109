"""Runs last (file order): how often did the fp64 arbiter of tests/parity.py decide a gradient
comparison in this session?  Printed, and bounded, so the escape hatch cannot become the norm."""
import pytest

from tests import parity as P


@pytest.mark.gpu
def test_fp64_arbiter_tally():
    checked, fp64 = P.ARBITER["checked"], P.ARBITER["fp64"]
    print(f"\n[parity] gradient tensors checked: {checked}; decided by the fp64 arbiter: {fp64} "
          f"(of which by the fp32 noise floor: {P.ARBITER['floor']})")
    for line in P.ARBITER["names"]:
        print("   arbiter:", line)
    ties = P.ARBITER.get("relu_ties", [])
    for line in ties:
        print("   relu tie:", line)
    if checked >= 100:       # only meaningful when the model tests ran in this process
        assert fp64 <= 0.05 * checked, f"fp64 arbiter decided {fp64} of {checked} gradient tensors"
        # rule (4) re-evaluates the oracle with ReLU decisions flipped: bounded too (each use covers one test's
        # ~14 tensors; 'tie-demo' is the pinned demonstration) -- more than 1 % of the checked tensors' tests
        # or more than 8 uses in a session means it has become a habit
        assert len(ties) <= max(8, 0.01 * checked / 14), f"ReLU-tie resolutions: {len(ties)} uses"
