"""Shared helpers for the parity tests (test infrastructure; may use the oracle)."""
import json
import os
import random

from oracle import bn254_ref as ref
from oracle import cpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "msm_vectors.json")
P, R = ref.P, ref.R


def golden_cases():
    with open(GOLDEN) as f:
        return json.load(f)["cases"]


GOLDEN_G2 = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "msm_vectors_g2.json")


def golden_cases_g2():
    """Known answers for G2 of BN254 / BLS12-381 (tests/golden/make_golden_g2.py); each case names its curve"""
    with open(GOLDEN_G2) as f:
        return json.load(f)["cases"]


def case_inputs_g2(case):
    import importlib

    if case["kind"] == "explicit":
        return bytes.fromhex(case["points"]), bytes.fromhex(case["scalars"])
    c = importlib.import_module("oracle.cpu_" + case["curve"])
    return c.sample_points(case["point_seed"], case["n"]), c.sample_scalars(case["scalar_seed"], case["n"])


def case_inputs(case):
    if case["kind"] == "explicit":
        return bytes.fromhex(case["points"]), bytes.fromhex(case["scalars"])
    return cpu.sample_points(case["point_seed"], case["n"]), cpu.sample_scalars(case["scalar_seed"], case["n"])


def b32(x):
    return int(x).to_bytes(32, "little")


def jacobian_bytes(pt, rnd):
    """Affine tuple (or None) -> 96 B Jacobian with a random z."""
    if pt is None:
        return bytes(96)
    z = rnd.randrange(1, P)
    return b32(pt[0] * z * z % P) + b32(pt[1] * z * z * z % P) + b32(z)


def affine64_list(xyz_bytes):
    return [cpu.to_affine64(xyz_bytes[i:i + 96]) for i in range(0, len(xyz_bytes), 96)]


def rng(seed):
    return random.Random(seed)
