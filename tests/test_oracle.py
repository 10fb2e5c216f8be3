"""Pin the CPU oracle: against every known-answer constant the reference holds for this path, against the independent
pure-Python model, and against the committed golden vectors.  (The reference has no MSM vectors: SURVEY.md 8c.)"""
import pytest

from oracle import bn254_ref as ref
from oracle import cpu
from tests.util import case_inputs, golden_cases

P, R = ref.P, ref.R


def test_reference_known_answer_limbs_13bit():
    # /root/reference/src/cuzk/utils.rs:439-451  to_words_le(0x12ab65..., 20, 13)
    v = 0x12AB655E9A2CA55660B44D1E5C37B00159AA76FED00000010A11800000000001
    assert ref.to_words_le(v, 20, 13) == [1, 0, 0, 768, 4257, 0, 0, 8154, 2678, 2765, 3072, 6255, 4581, 6694, 6530, 5290, 6700,
                                          2804, 2777, 37]


def test_reference_known_answer_moduli_16bit_limbs():
    # /root/reference/src/naive/wgsl/bn254/field.wgsl:5 and src/naive/utils/bigint.rs:92 : p as 16 x u16 little-endian
    assert ref.to_words_le(P, 16, 16) == [64839, 55420, 35862, 15392, 51853, 26737, 27281, 38785, 22621, 33153, 17846, 47184,
                                          41001, 57649, 20082, 12388]
    # MONTGOMERY_INV = 25481 = -p^-1 mod 2^16 (src/naive/wgsl/bn254/field.wgsl:25)
    assert (-pow(P, -1, 1 << 16)) % (1 << 16) == 25481
    # p as 20 x 13-bit limbs and n0 = 905 for the reference's own representation (utils.rs:339-373)
    assert ref.to_words_le(P, 20, 13)[:4] == [7495, 999, 1462, 280]
    assert (-pow(P, -1, 1 << 13)) % (1 << 13) == 905
    assert P == int("21888242871839275222246405745257275088696311157297823662689037894645226208583")  # msm.rs:39


def test_oracle_constants():
    c = cpu.constants()
    assert c["p"] == P and c["r"] == R
    assert c["R_mod_p"] == (1 << 256) % P == 0x0E0A77C19A07DF2F666EA36F7879462C0A78EB28F5C70B3DD35D438DC58F0D9D
    assert c["R2_mod_p"] == pow(2, 512, P) == 0x06D89F71CAB8351F47AB1EFF0A417FF6B5E71911D44501FBF32CFC5B538AFA89
    assert c["n0inv64"] == 0x87D20782E4866389 and (c["n0inv64"] * P + 1) % (1 << 64) == 0
    assert (R >> 240) == 0x3064 and (R >> 240) < (1 << 15)  # why the recode never carries out (test/utils.rs:149-152)


def test_public_known_answers():
    G = ref.points_to_bytes([ref.G])
    two = cpu.to_affine64(cpu.g1_scalar_mul(G, (2).to_bytes(32, "little")))
    assert int.from_bytes(two[:32], "little") == 0x030644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD3
    assert int.from_bytes(two[32:], "little") == 0x15ED738C0E0A7C92E7845F96B2AE9C0A68A6A449E3538FC7FF3EBF7A5A18A2C4
    assert cpu.to_affine64(cpu.g1_scalar_mul(G, (R - 1).to_bytes(32, "little"))) == ref.affine_to_bytes64((1, P - 2))
    assert cpu.to_affine64(cpu.g1_scalar_mul(G, R.to_bytes(32, "little"))) == bytes(64)


def test_samplers_agree_and_points_on_curve():
    assert cpu.sample_scalars(7, 50) == ref.scalars_to_bytes(ref.sample_scalars(7, 50))
    pts = ref.sample_points(7, 20)
    assert cpu.sample_points(7, 20) == ref.points_to_bytes(pts)
    assert all(ref.is_on_curve(p) for p in pts) and cpu.points_on_curve(cpu.sample_points(9, 500))
    assert cpu.sample_points(7, 5, first=10) == ref.points_to_bytes(pts[10:15])


@pytest.mark.parametrize("n", [1, 2, 3, 5, 31, 32, 33, 100])
def test_c_msm_equals_python_definition(n):
    pts, sc = ref.sample_points(11, n), ref.sample_scalars(13, n)
    pb, sb = ref.points_to_bytes(pts), ref.scalars_to_bytes(sc)
    want = ref.affine_to_bytes64(ref.msm_naive(pts, sc))
    assert cpu.to_affine64(cpu.cpu_msm(pb, sb)) == want
    assert cpu.to_affine64(cpu.cpu_msm(pb, sb, n_threads=3)) == want
    assert cpu.to_affine64(cpu.msm_cuzk_model(pb, sb)) == want


def test_stage_models_c_vs_python_small_window():
    # c = 8 keeps the pure-Python stage models fast; every stage is compared, then the pipeline result
    n, c = 300, 8
    pts, sc = ref.sample_points(17, n), ref.sample_scalars(19, n)
    pb, sb = ref.points_to_bytes(pts), ref.scalars_to_bytes(sc)
    dig_py = ref.decompose_scalars_signed(sc, 32, c)
    dig_c = cpu.decompose_scalars_signed(sb, 32, c)
    assert dig_c.tolist() == dig_py
    for w in (0, 13, 31):
        cp_py, val_py = ref.cpu_transpose(dig_py[w], 1 << c)
        cp_c, val_c = cpu.transpose(dig_c[w], 1 << c)
        assert cp_c.tolist() == cp_py and val_c.tolist() == val_py
        b_py = ref.cpu_smvp_signed(cp_py, val_py, pts, 1 << c)
        b_c = cpu.smvp_signed(cp_c, val_c, pb, 1 << c)
        assert [cpu.to_affine64(b_c[96 * k:96 * k + 96]) for k in range(1 << (c - 1))] == [ref.affine_to_bytes64(b) for b in b_py]
        want = ref.affine_to_bytes64(ref.serial_bucket_reduction(b_py))
        assert ref.affine_to_bytes64(ref.running_sum_bucket_reduction(b_py)) == want
        acc = None
        for g in ref.parallel_bucket_reduction(b_py, 4):
            acc = ref.add(acc, g)
        assert ref.affine_to_bytes64(acc) == want  # ≙ tests/cuzk.rs:60-61,75-76
        for kind, nt in (("serial", 1), ("running_sum", 1), ("parallel", 4), ("parallel", 16)):
            assert cpu.to_affine64(cpu.bucket_reduction(kind, b_c, nt)) == want
        g1, m1 = cpu.parallel_bucket_reduction_1(b_c, 8)
        two = cpu.parallel_bucket_reduction_2(g1, m1, 1 << (c - 1), 8)
        acc = bytes(96)
        for t in range(8):
            acc = cpu.g1_op("add", acc, two[96 * t:96 * t + 96])
        assert cpu.to_affine64(acc) == want
    assert ref.affine_to_bytes64(ref.msm_cuzk_model(pts, sc, c)) == ref.affine_to_bytes64(ref.msm_naive(pts, sc))
    assert cpu.to_affine64(cpu.msm_cuzk_model(pb, sb, c)) == ref.affine_to_bytes64(ref.msm_naive(pts, sc))


def test_recode_edge_scalars():
    for s in (0, 1, 0x7FFF, 0x8000, 0xFFFF, R - 1, int("8000" * 15, 16), (1 << 250) - 1):
        d = cpu.decompose_scalars_signed(s.to_bytes(32, "little"))[:, 0].astype(int) - (1 << 15)
        assert sum(int(x) << (16 * w) for w, x in enumerate(d)) == s
        assert all(-(1 << 15) <= x < (1 << 15) for x in d)
    with pytest.raises(ValueError):
        cpu.decompose_scalars_signed(b"\xff" * 32)  # final carry (test/utils.rs:150-152)


@pytest.mark.parametrize("case", golden_cases(), ids=lambda c: c["name"])
def test_oracle_reproduces_golden(case):
    points, scalars = case_inputs(case)
    assert cpu.to_affine64(cpu.cpu_msm(points, scalars)).hex() == case["expected_affine"]
    if len(scalars) // 32 <= 5000:
        assert cpu.to_affine64(cpu.msm_cuzk_model(points, scalars)).hex() == case["expected_affine"]


def test_reference_known_answers_pasta():
    """The Pallas constants the reference holds (its dead second curve): 16-bit limbs of p, BASE_M and U in
    /root/reference/src/naive/wgsl/pallas/field.wgsl:4-6,18-24, the decimal p, q, base_m, u and the identity p = 2^254 + u in
    src/naive/utils/bigint.rs:38-75, the hex modulus in src/naive/utils/files.rs:26 -- both Pasta oracles are pinned to them."""
    from oracle import cpu_pallas, cpu_vesta, pallas_ref, vesta_ref

    fp = int("28948022309329048855892746252171976963363056481941560715954676764349967630337")   # bigint.rs:38
    fq = int("28948022309329048855892746252171976963363056481941647379679742748393362948097")   # bigint.rs:39
    base_m = int("115792089237316195423570985008687907853087743403514885215096460958426388758524")  # bigint.rs:40
    u = int("45560315531419706090280762371685220353")                                           # bigint.rs:59
    assert fp == (1 << 254) + u and (4 * ((1 << 254) - u)) % (1 << 256) == base_m                # bigint.rs:60-64
    assert fp == 0x40000000000000000000000000000000224698FC094CF91B992D30ED00000001             # files.rs:26
    assert ref.to_words_le(fp, 16, 16) == [1, 0, 12525, 39213, 63771, 2380, 39164, 8774, 0, 0, 0, 0, 0, 0, 0, 16384]   # field.wgsl:5, bigint.rs:72
    assert ref.to_words_le(fq, 16, 16) == [1, 0, 60193, 35910, 43229, 2452, 39164, 8774, 0, 0, 0, 0, 0, 0, 0, 16384]   # bigint.rs:73
    assert ref.to_words_le(base_m, 16, 16) == [65532, 65535, 15435, 39755, 7057, 56012, 39951, 30437] + [65535] * 8    # field.wgsl:18-20, bigint.rs:74
    assert ref.to_words_le(u, 16, 16) == [1, 0, 12525, 39213, 63771, 2380, 39164, 8774, 0, 0, 0, 0, 0, 0, 0, 0]        # field.wgsl:22-24, bigint.rs:75
    # the moduli both oracle models and both C builds actually compute with
    assert (pallas_ref.P, pallas_ref.R) == (fp, fq) and (vesta_ref.P, vesta_ref.R) == (fq, fp)
    cp, cv = cpu_pallas.constants(), cpu_vesta.constants()
    assert (cp["p"], cp["r"]) == (fp, fq) and (cv["p"], cv["r"]) == (fq, fp)
    assert cp["R_mod_p"] == (1 << 256) % fp and cv["R_mod_p"] == (1 << 256) % fq
    # the group orders: q G = infinity on Pallas, p G = infinity on Vesta (generator (-1, 2)), and (order - 1) G = -G
    for cpu_c, model, order in ((cpu_pallas, pallas_ref, fq), (cpu_vesta, vesta_ref, fp)):
        G = model.points_to_bytes([model.G])
        assert cpu_c.to_affine64(cpu_c.g1_scalar_mul(G, order.to_bytes(32, "little"))) == bytes(64)
        assert cpu_c.to_affine64(cpu_c.g1_scalar_mul(G, (order - 1).to_bytes(32, "little"))) == model.affine_to_bytes64((model.P - 1, model.P - 2))
