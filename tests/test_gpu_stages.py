"""Per-stage parity with the CPU stage models (≙ tests/decompose_shader.rs, tests/transpose_shader.rs:172-200,
tests/smvp_shader.rs:292-334 of the reference, which compare each shader with src/cuzk/test/utils.rs).
The engine's CSC is keyed by bucket slot with the sign in bit 31, so the transpose check maps the
reference's rows h+k / h-k onto slot k; bucket sums and window sums are compared as group elements, bit-exact in the
canonical affine encoding."""
import numpy as np
import pytest

from oracle import cpu
from tests.util import affine64_list

pytestmark = pytest.mark.gpu

N = 70000  # > 2^16 so every window sees multi-entry buckets


@pytest.fixture(scope="module", params=[16, 14, 12], ids=lambda b: "c%d" % b)
def run(ctx, request):
    # every window size the engine supports (the reference's is 16, src/cuzk/msm.rs:79); the CPU stage models take the window
    # size as a parameter (src/cuzk/test/utils.rs:121 `word_size`)
    bits = request.param
    W, H = ctx.window_config(bits)
    points, scalars = cpu.sample_points(60, N), cpu.sample_scalars(61, N)
    # adversarial rows: digit -2^(c-1) (slot 0), zero scalar, duplicates
    sc = bytearray(scalars)
    sc[0:32] = (H).to_bytes(32, "little")
    sc[32:64] = bytes(32)
    sc[64:96] = sc[96:128]
    pt = bytearray(points)
    pt[64 * 2:64 * 3] = pt[64 * 3:64 * 4]
    points, scalars = bytes(pt), bytes(sc)
    ctx.set_bases(points)
    ctx.set_debug(True)
    ctx.set_window_bits(bits)
    try:
        result = ctx.msm(scalars)
        assert ctx.last_window_bits() == bits
    finally:
        ctx.set_debug(False)
        ctx.set_window_bits(0)
    return {"bits": bits, "W": W, "H": H, "points": points, "scalars": scalars, "result": result, "digits": ctx.read_digits(N, W),
            "col_ptr": ctx.read_col_ptr(W, H), "val": ctx.read_val_idxs(N, W), "buckets": ctx.read_buckets(W, H),
            "wsums": ctx.read_window_sums(W), "model_digits": cpu.decompose_scalars_signed(scalars, W, bits)}


def test_decompose_matches_cpu_model(run):
    H, W, bits = run["H"], run["W"], run["bits"]
    biased = run["model_digits"].astype(np.int64)  # d + 2^(c-1)
    d = biased - H
    code = run["digits"].astype(np.int64)
    mag = code & 0x7FFF
    sign = code >> 15
    got = np.where(sign == 1, -np.where(mag == 0, H, mag), mag)
    assert np.array_equal(got, d)
    # every scalar is reassembled exactly from its digits: s = sum_w d_w 2^(c w)
    s0 = int.from_bytes(run["scalars"][96:128], "little")
    assert sum(int(got[w, 3]) << (bits * w) for w in range(W)) == s0


def test_transpose_matches_cpu_model_rows(run):
    col_ptr, val, H, W = run["col_ptr"], run["val"], run["H"], run["W"]
    for w in (0, 7, W - 1):
        ref_cp, ref_val = cpu.transpose(run["model_digits"][w], 2 * H)
        assert col_ptr[w, 0] == 0 and np.all(np.diff(col_ptr[w].astype(np.int64)) >= 0)
        nz = int(np.count_nonzero(run["model_digits"][w] != H))
        assert col_ptr[w, H] == nz
        for k in list(range(0, 40)) + [12345 % H, H - 1]:
            got = val[w, col_ptr[w, k]:col_ptr[w, k + 1]]
            pos = set() if k == 0 else set(ref_val[ref_cp[H + k]:ref_cp[H + k + 1]].tolist())
            neg_row = 0 if k == 0 else H - k
            neg = set(ref_val[ref_cp[neg_row]:ref_cp[neg_row + 1]].tolist())
            assert {int(v) for v in got if not (v >> 31)} == pos
            assert {int(v & 0x7FFFFFFF) for v in got if v >> 31} == neg
            # ... and, in the debug mode this fixture runs in, in a DETERMINISTIC order (k_order_runs): the reference's stage test asserts
            # the exact val_idxs (tests/transpose_shader.rs:198-199); here a slot is the reference's row h + k (ascending point index, as
            # its serial loop leaves it) followed by its row h - k with bit 31 set
            ref_pos = [] if k == 0 else ref_val[ref_cp[H + k]:ref_cp[H + k + 1]].tolist()
            ref_neg = ref_val[ref_cp[neg_row]:ref_cp[neg_row + 1]].tolist()
            assert ref_pos == sorted(ref_pos) and ref_neg == sorted(ref_neg)
            assert [int(v) for v in got] == ref_pos + [int(v) | 0x80000000 for v in ref_neg]


def test_smvp_buckets_match_cpu_model(run):
    H, W = run["H"], run["W"]
    for w in (0, 9, W - 1):
        cp, vi = cpu.transpose(run["model_digits"][w], 2 * H)
        want = cpu.smvp_signed(cp, vi, run["points"], 2 * H)
        got = run["buckets"][w].tobytes()
        assert affine64_list(got) == affine64_list(want)


def test_bucket_reduction_matches_cpu_models(run):
    for w in (0, 5, run["W"] - 1):
        buckets = run["buckets"][w].tobytes()
        want = cpu.to_affine64(cpu.bucket_reduction("running_sum", buckets))
        assert cpu.to_affine64(run["wsums"][w].tobytes()) == want
    # the reference's own split (256 simulated threads, two stages) agrees with the running sum
    b0 = run["buckets"][0].tobytes()
    assert cpu.to_affine64(cpu.bucket_reduction("parallel", b0, 256)) == cpu.to_affine64(run["wsums"][0].tobytes())


def test_horner_of_window_sums_is_the_result(run):
    assert cpu.to_affine64(cpu.horner(run["wsums"].tobytes(), run["bits"])) == run["result"].to_affine_bytes()
    assert run["result"].to_affine_bytes() == cpu.to_affine64(cpu.cpu_msm(run["points"], run["scalars"]))


def test_window_sums_read_back_after_a_plain_launch(ctx):
    """Without debug a single-MSM launch leaves the bucket reduce's bit-plane sums on the device (k_bpr_planes: the host finishes
    the window sums); msm_hip_read_window_sums turns them into the documented 96-byte records."""
    n = 5000
    points, scalars = cpu.sample_points(62, n), cpu.sample_scalars(63, n)
    ctx.set_bases(points)
    ctx.set_window_bits(16)
    try:
        result = ctx.msm(scalars)
        wsums = ctx.read_window_sums(16)
    finally:
        ctx.set_window_bits(0)
    assert cpu.to_affine64(cpu.horner(wsums.tobytes(), 16)) == result.to_affine_bytes()
    assert result.to_affine_bytes() == cpu.to_affine64(cpu.cpu_msm(points, scalars))
