"""Randomised differential test against the oracle (sizes on tile / chunk / alignment edges, skewed and edge-case scalar
distributions, duplicate and negated points, host / device / window-sharded / batch entry points): tools/fuzz_gpu.py."""
import os
import runpy
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fuzz_against_oracle(built, capsys):
    argv = sys.argv
    sys.argv = ["fuzz_gpu.py", "60", "20261004"]
    try:
        runpy.run_path(os.path.join(ROOT, "tools", "fuzz_gpu.py"), run_name="__main__")
    finally:
        sys.argv = argv
    assert "fuzz ok: 60 cases" in capsys.readouterr().out


@pytest.mark.parametrize("curve,cases", [("grumpkin", 30), ("pallas", 12), ("vesta", 12), ("bls12_381", 8), ("bn254_g2", 10), ("bls12_381_g2", 6)])
def test_fuzz_against_oracle_other_curves(built, capsys, curve, cases):
    argv = sys.argv
    sys.argv = ["fuzz_gpu.py", str(cases), "20261005", curve]
    try:
        runpy.run_path(os.path.join(ROOT, "tools", "fuzz_gpu.py"), run_name="__main__")
    finally:
        sys.argv = argv
    assert "fuzz ok: %d cases" % cases in capsys.readouterr().out


def test_fuzz_wide_tables_against_oracle(built, capsys, monkeypatch):
    """the wide fixed-base tables only (whole MSMs, batches, shares of the virtual windows, the multi-GPU calls), at every digit width"""
    monkeypatch.setenv("FUZZ_MODES", "wide,wide_batch,wide_shares,mgpu_wide")
    argv = sys.argv
    sys.argv = ["fuzz_gpu.py", "40", "20261006"]
    try:
        runpy.run_path(os.path.join(ROOT, "tools", "fuzz_gpu.py"), run_name="__main__")
    finally:
        sys.argv = argv
    assert "fuzz ok: 40 cases" in capsys.readouterr().out


@pytest.mark.parametrize("curve,cases", [("bn254", 24), ("grumpkin", 10), ("bls12_381", 6), ("bn254_g2", 6)])
def test_fuzz_upload_bound_calls_in_parts(built, capsys, monkeypatch, curve, cases):
    """round 5: the one-shot call and msm_hip_run with host scalars as sums of 1 - 4 sub-MSMs over ranges of the points"""
    monkeypatch.setenv("FUZZ_MODES", "oneshot_parts,run_parts")
    argv = sys.argv
    sys.argv = ["fuzz_gpu.py", str(cases), "20261007", curve]
    try:
        runpy.run_path(os.path.join(ROOT, "tools", "fuzz_gpu.py"), run_name="__main__")
    finally:
        sys.argv = argv
    assert "fuzz ok: %d cases" % cases in capsys.readouterr().out
