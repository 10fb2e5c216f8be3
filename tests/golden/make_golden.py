#!/usr/bin/env python3
"""Generate tests/golden/msm_vectors.json.

The reference holds no MSM vectors and cannot run here, so these vectors are produced by the
independent pure-Python big-integer model (oracle/bn254_ref.py, msm_naive = sum of double-and-add products) and
cross-checked against the C restatement (oracle/bn254.c) before being written.  Small cases carry explicit inputs;
larger ones carry the sampler seeds (the sampler is defined identically in bn254_ref.py, bn254.c and the HIP kernels).
Run from the repo root:  python tests/golden/make_golden.py
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import bn254_ref as ref  # noqa: E402
from oracle import cpu  # noqa: E402

R, P, G = ref.R, ref.P, ref.G


def case_explicit(name, points, scalars, note=""):
    want = ref.affine_to_bytes64(ref.msm_naive(points, scalars))
    pb, sb = ref.points_to_bytes(points), ref.scalars_to_bytes(scalars)
    assert cpu.to_affine64(cpu.cpu_msm(pb, sb)) == want, name
    assert cpu.to_affine64(cpu.msm_cuzk_model(pb, sb)) == want, name
    return {"name": name, "kind": "explicit", "points": pb.hex(), "scalars": sb.hex(), "expected_affine": want.hex(), "note": note}


def case_seeded(name, n, pseed, sseed, python_check):
    pb, sb = cpu.sample_points(pseed, n), cpu.sample_scalars(sseed, n)
    want = cpu.to_affine64(cpu.cpu_msm(pb, sb))
    if python_check:
        pts, sc = ref.sample_points(pseed, n), ref.sample_scalars(sseed, n)
        assert ref.points_to_bytes(pts) == pb and ref.scalars_to_bytes(sc) == sb
        assert ref.affine_to_bytes64(ref.msm_naive(pts, sc)) == want, name
    assert cpu.to_affine64(cpu.msm_cuzk_model(pb, sb)) == want, name
    return {"name": name, "kind": "seeded", "n": n, "point_seed": pseed, "scalar_seed": sseed, "expected_affine": want.hex(),
            "checked_by": "python+c" if python_check else "c(two algorithms)"}


def main():
    pts = ref.sample_points(1001, 16)
    cases = []
    cases.append(case_explicit("generator_times_2", [G], [2], "public KAT: 2G (SURVEY.md Appendix B)"))
    cases.append(case_explicit("generator_times_r_minus_1", [G], [R - 1], "(r-1)G = (1, p-2)"))
    cases.append(case_explicit("n1_random", [pts[0]], [ref.sample_scalar(5, 0)]))
    cases.append(case_explicit("n2_cancel", [pts[1], pts[1]], [7, R - 7], "s*P + (-s)*P = identity"))
    cases.append(case_explicit("n3", pts[:3], ref.sample_scalars(6, 3)))
    cases.append(case_explicit("zero_scalars", pts[:4], [0, 0, 0, 0], "all digits zero -> identity"))
    cases.append(case_explicit("one_nonzero", pts[:4], [0, 0, 12345, 0]))
    cases.append(case_explicit("digit_0x8000_chain", pts[:2], [int("8000" * 15, 16) + (0x1000 << 240), 0x8000],
                               "every 16-bit digit = 0x8000 -> recode to -2^15 with carries (bucket slot 0)"))
    cases.append(case_explicit("digit_0xffff_chain", pts[:2], [(1 << 250) - 1, 0xFFFF], "digits 0xffff: carry ripples through zero digits"))
    cases.append(case_explicit("duplicate_points_same_bucket", [pts[2]] * 5, [3, 3, 3, 3, 3], "P = Q inside one bucket -> doubling path"))
    cases.append(case_explicit("p_and_minus_p_same_bucket", [pts[3], ref.neg(pts[3]), pts[4]], [9, 9, 1], "P + (-P) -> identity path"))
    cases.append(case_explicit("same_point_opposite_digits", [pts[5], pts[5]], [5, R - 5], "digit +5 and the recode of r-5"))
    cases.append(case_explicit("all_equal_scalars", pts[:16], [0xABCDEF0123456789ABCDEF0123456789] * 16, "single-bucket skew"))
    cases.append(case_explicit("small_scalars", pts[:8], list(range(1, 9))))
    cases.append(case_explicit("r_minus_1_many", pts[:6], [R - 1] * 6))
    cases.append(case_seeded("seeded_n17", 17, 21, 22, True))
    cases.append(case_seeded("seeded_n256", 256, 23, 24, True))
    cases.append(case_seeded("seeded_n1000", 1000, 25, 26, False))
    cases.append(case_seeded("seeded_n4096", 4096, 27, 28, False))
    cases.append(case_seeded("seeded_n65540", (1 << 16) + 4, 29, 30, False))
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "msm_vectors.json")
    with open(out, "w") as f:
        json.dump({"format": "points n x 64 B (x||y LE), scalars n x 32 B LE, expected = 64 B affine LE (zeros = identity)",
                   "cases": cases}, f, indent=1)
    print("wrote", out, len(cases), "cases")


if __name__ == "__main__":
    main()
