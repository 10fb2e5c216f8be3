#!/usr/bin/env python3
"""Generate tests/golden/msm_vectors_g2.json: known answers for MSMs on G2 of BN254 and of BLS12-381.

The reference has no G2 code; these vectors are produced by the pure-Python model (oracle/bn254_g2_ref.py and its BLS12-381 instance:
msm_naive = sum of double-and-add products) and cross-checked, before being written, against the G2 build of the C restatement
(oracle/bn254.c -DORACLE_G2: MSM and the cuZK stage models) and against the closed form over the points' known multipliers.
Small cases carry explicit inputs; larger ones the sampler seeds (the sampler -- known multiples of the generator -- is defined identically
in the Python and the C model).  Run from the repo root:  python tests/golden/make_golden_g2.py
"""
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def cases_for(curve):
    ref = importlib.import_module("oracle." + curve + "_ref")
    cpu = importlib.import_module("oracle.cpu_" + curve)
    R, G = ref.R, ref.G

    def explicit(name, points, scalars, note=""):
        want = ref.affine_to_bytes(ref.msm_naive(points, scalars))
        pb, sb = ref.points_to_bytes(points), ref.scalars_to_bytes(scalars)
        assert cpu.to_affine64(cpu.cpu_msm(pb, sb)) == want == cpu.to_affine64(cpu.msm_cuzk_model(pb, sb)), name
        return {"curve": curve, "name": name, "kind": "explicit", "points": pb.hex(), "scalars": sb.hex(), "expected_affine": want.hex(), "note": note}

    def seeded(name, n, pseed, sseed):
        pb, sb = cpu.sample_points(pseed, n), cpu.sample_scalars(sseed, n)
        want = cpu.to_affine64(cpu.cpu_msm(pb, sb))
        sc = ref.bytes_to_scalars(sb)
        assert ref.points_to_bytes(ref.sample_points(min(n, 64), pseed)) == pb[: min(n, 64) * 2 * ref.CB]
        assert ref.affine_to_bytes(ref.msm_by_multipliers(ref.sample_multipliers(n, pseed), sc)) == want, name
        if n <= 300:
            assert ref.affine_to_bytes(ref.msm_naive(ref.bytes_to_points(pb), sc)) == want, name
        assert cpu.to_affine64(cpu.msm_cuzk_model(pb, sb)) == want, name
        return {"curve": curve, "name": name, "kind": "seeded", "n": n, "point_seed": pseed, "scalar_seed": sseed, "expected_affine": want.hex(),
                "checked_by": "python model (closed form%s) + c (two algorithms)" % (", double-and-add" if n <= 300 else "")}

    pts = ref.sample_points(16, 1001)
    return [
        explicit("generator_times_2", [G], [2], "2 G of the public generator"),
        explicit("generator_times_r_minus_1", [G], [R - 1], "(r - 1) G = -G"),
        explicit("n2_cancel", [pts[1], pts[1]], [7, R - 7], "s P + (-s) P = identity"),
        explicit("n3", pts[:3], [ref.sample_scalar(6, i) for i in range(3)]),
        explicit("zero_scalars", pts[:4], [0, 0, 0, 0], "all digits zero -> identity"),
        explicit("digit_0x8000_chain", pts[:2], [int("8000" * 15, 16) + (0x1000 << 240), 0x8000], "every 16-bit digit = 0x8000: recode to -2^15 with carries (bucket slot 0)"),
        explicit("duplicate_points_same_bucket", [pts[2]] * 5, [3] * 5, "P = Q inside one bucket: doubling path"),
        explicit("p_and_minus_p_same_bucket", [pts[3], ref.neg(pts[3]), pts[4]], [9, 9, 1], "P + (-P): identity path"),
        explicit("all_equal_scalars", pts[:16], [0xABCDEF0123456789ABCDEF0123456789] * 16, "single-bucket skew"),
        explicit("r_minus_1_many", pts[:6], [R - 1] * 6),
        seeded("seeded_n17", 17, 21, 22),
        seeded("seeded_n256", 256, 23, 24),
        seeded("seeded_n4096", 4096, 27, 28),
        seeded("seeded_n65540", (1 << 16) + 4, 29, 30),
    ]


def main():
    cases = cases_for("bn254_g2") + cases_for("bls12_381_g2")
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "msm_vectors_g2.json")
    with open(out, "w") as f:
        json.dump({"format": "coordinates are Fq2 elements c0 || c1 (2 x 32 B little-endian; 2 x 48 B on bls12_381_g2); points n x (x || y), scalars n x 32 B LE, "
                             "expected = affine x || y (zeros = identity)", "cases": cases}, f, indent=1)
    print("wrote", out, len(cases), "cases")


if __name__ == "__main__":
    main()
