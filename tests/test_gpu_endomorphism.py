"""The curve-endomorphism mode ("GLV endomorphism for BN254 to halve scalar length"; the reference uses
full-length scalars, src/cuzk/msm.rs:79-82): MSM_HIP_BASES_ENDOMORPHISM stores phi(P_i) behind the bases, every scalar is split
k = k1 + k2 lambda on the device, and the MSM runs over the 2n points with half the windows.

Checked stage by stage with the reference's CPU stage models (src/cuzk/test/utils.rs:61-338, restated in oracle/) applied to the
EQUIVALENT full-sign problem -- points sgn(k1_i) P_i, sgn(k2_i) phi(P_i), scalars |k1_i|, |k2_i| from the oracle's own split -- and
end to end against the oracle's plain MSM, for every window size and both curves."""
import numpy as np
import pytest
import torch

import msm_webgpu_amd as m
from oracle import bn254_ref as ref
from oracle import cpu
from tests.util import affine64_list, b32

pytestmark = pytest.mark.gpu
N = 40000


def equivalent_problem(model, points, scalars):
    """-> (2n points with the halves' signs applied, 2n magnitudes as 32-byte scalars, [neg flag per input])"""
    pts = model.bytes_to_points(points)
    ks = model.bytes_to_scalars(scalars)
    halves = [model.glv_split(k) for k in ks]
    out_p, out_s, negs = [], [], []
    for j in range(2):
        for pt, h in zip(pts, halves):
            q = pt if j == 0 else model.endo(pt)
            negs.append(1 if h[j] < 0 else 0)
            out_p.append(model.neg(q) if h[j] < 0 else q)
            out_s.append(abs(h[j]))
    return model.points_to_bytes(out_p), b"".join(b32(v) for v in out_s), negs


def _halves_first_then_second(planes):
    """The device numbers its 2n inputs interleaved (input 2 j = first half of scalar j, 2 j + 1 = its second half: the first sort pass
    splits a scalar and handles both halves); the equivalent problem lists all first halves, then all second halves."""
    return np.concatenate([planes[:, 0::2], planes[:, 1::2]], axis=1)


@pytest.fixture(scope="module", params=[16, 14, 12], ids=lambda b: "c%d" % b)
def run(ctx, request):
    bits = request.param
    W, H = ctx.endomorphism_window_count(bits), 1 << (bits - 1)
    points, scalars = cpu.sample_points(90, N), bytearray(cpu.sample_scalars(91, N))
    lam = ref.glv_params()["lam"]
    # adversarial rows: 0, 1, r - 1, lambda (k1 = 0, k2 = 1), r - lambda, a scalar whose first half has the digit -2^(c-1)
    # (which a negative half turns into +2^(c-1): slot 0 with a positive sign), duplicates
    special = [0, 1, ref.R - 1, lam, ref.R - lam, H, (H * lam) % ref.R, ref.R - H, (ref.R - H * lam) % ref.R]
    for i, v in enumerate(special):
        scalars[32 * i:32 * i + 32] = b32(v)
    scalars[32 * 20:32 * 21] = scalars[32 * 21:32 * 22]
    scalars = bytes(scalars)
    ctx.set_bases(points, endomorphism=True)
    ctx.set_debug(True)
    ctx.set_window_bits(bits)
    try:
        result = ctx.msm(scalars)
        assert ctx.last_window_bits() == bits
    finally:
        ctx.set_debug(False)
        ctx.set_window_bits(0)
    eq_points, eq_scalars, negs = equivalent_problem(ref, points, scalars)
    return {"bits": bits, "W": W, "H": H, "points": points, "scalars": scalars, "result": result, "negs": np.array(negs, dtype=np.int64),
            "eq_points": eq_points, "eq_scalars": eq_scalars, "digits": _halves_first_then_second(ctx.read_digits(2 * N, W)), "col_ptr": ctx.read_col_ptr(W, H),
            "val": ctx.read_val_idxs(2 * N, W), "buckets": ctx.read_buckets(W, H), "wsums": ctx.read_window_sums(W),
            "model_digits": cpu.decompose_scalars_signed(eq_scalars, W, bits)}


def test_split_and_recode_match_the_model(run):
    H, W, bits = run["H"], run["W"], run["bits"]
    d = run["model_digits"].astype(np.int64) - H                 # signed digits of the magnitudes
    want = np.where(run["negs"][None, :] == 1, -d, d)              # ... with the half's sign applied
    code = run["digits"].astype(np.int64)
    mag, sign = code & 0x7FFF, code >> 15
    got = np.where(sign == 1, -np.where(mag == 0, H, mag), mag)
    # the debug plane's 15-bit magnitude cannot hold +2^(c-1) (a negative half's digit -2^(c-1)); it shows 0 there, and the
    # transpose test below finds the entry in slot 0 with a positive sign
    plus_h = want == H
    assert np.array_equal(np.where(plus_h, 0, want), got)
    assert plus_h.sum() > 0, "the adversarial rows were meant to produce such a digit"
    # the halves reassemble every scalar: k = k1 + k2 lambda (mod r)
    lam = ref.glv_params()["lam"]
    ks = ref.bytes_to_scalars(run["scalars"])
    for i in list(range(12)) + [N - 1]:
        k1 = sum(int(want[w, i]) << (bits * w) for w in range(W))
        k2 = sum(int(want[w, N + i]) << (bits * w) for w in range(W))
        assert (k1 + k2 * lam - ks[i]) % ref.R == 0
        assert abs(k1) < 1 << 127 and abs(k2) < 1 << 127


def test_transpose_matches_cpu_model_rows(run):
    col_ptr, val, H, W, negs = run["col_ptr"], run["val"], run["H"], run["W"], run["negs"]
    for w in (0, W // 2, W - 1):
        ref_cp, ref_val = cpu.transpose(run["model_digits"][w], 2 * H)
        assert col_ptr[w, 0] == 0 and np.all(np.diff(col_ptr[w].astype(np.int64)) >= 0)
        assert col_ptr[w, H] == int(np.count_nonzero(run["model_digits"][w] != H))
        for k in list(range(0, 30)) + [H - 1]:
            got = val[w, col_ptr[w, k]:col_ptr[w, k + 1]]
            pos = set() if k == 0 else set(ref_val[ref_cp[H + k]:ref_cp[H + k + 1]].tolist())
            neg_row = 0 if k == 0 else H - k
            neg = set(ref_val[ref_cp[neg_row]:ref_cp[neg_row + 1]].tolist())
            # an entry's sign is the digit's sign XOR the half's sign
            got_pos = {int(v & 0x7FFFFFFF) for v in got if (int(v) >> 31) ^ int(negs[int(v & 0x7FFFFFFF)]) == 0}
            got_neg = {int(v & 0x7FFFFFFF) for v in got if (int(v) >> 31) ^ int(negs[int(v & 0x7FFFFFFF)]) == 1}
            assert got_pos == pos and got_neg == neg, (w, k)


def test_smvp_buckets_match_cpu_model(run):
    H, W = run["H"], run["W"]
    for w in (0, W - 1):
        cp, vi = cpu.transpose(run["model_digits"][w], 2 * H)
        want = cpu.smvp_signed(cp, vi, run["eq_points"], 2 * H)
        assert affine64_list(run["buckets"][w].tobytes()) == affine64_list(want)


def test_bucket_reduction_and_horner(run):
    for w in (0, run["W"] - 1):
        want = cpu.to_affine64(cpu.bucket_reduction("running_sum", run["buckets"][w].tobytes()))
        assert cpu.to_affine64(run["wsums"][w].tobytes()) == want
    assert cpu.to_affine64(cpu.horner(run["wsums"].tobytes(), run["bits"])) == run["result"].to_affine_bytes()
    assert run["result"].to_affine_bytes() == cpu.to_affine64(cpu.cpu_msm(run["points"], run["scalars"]))
    # and the equivalent problem is the same group element (the split itself, end to end on the CPU)
    assert run["result"].to_affine_bytes() == cpu.to_affine64(cpu.cpu_msm(run["eq_points"], run["eq_scalars"]))


@pytest.mark.parametrize("n,nb", [(1, 1), (2, 3), (63, 64), (257, 300), (4097, 4097), (65540, 65540)])
def test_end_to_end_all_entry_points(ctx, n, nb):
    points, sc = cpu.sample_points(92, nb), cpu.sample_scalars(93, n)
    want = cpu.to_affine64(cpu.cpu_msm(points[:64 * n], sc))
    ctx.set_bases(points, endomorphism=True)
    assert ctx.uses_endomorphism()
    dev = torch.frombuffer(bytearray(sc), dtype=torch.uint8).cuda()
    assert ctx.msm(sc).to_affine_bytes() == want
    assert ctx.msm(dev).to_affine_bytes() == want
    for slot in range(3):
        ctx.launch_host(sc, slot)
    assert all(ctx.finish(slot).to_affine_bytes() == want for slot in range(3))
    assert [g.to_affine_bytes() for g in ctx.msm_batch(sc * 9, n)] == [want] * 9   # 8 MSMs per launch + 1
    # window-sharded launches use the plain bases (records 0 .. n-1) and the reference's 16 windows
    parts = [ctx.msm_windows(dev, 0, 5), ctx.msm_windows(dev, 5, 16)]
    assert m.MsmContext.combine_windows(torch.cat(parts, dim=0)).to_affine_bytes() == want
    # scalars in R = 2^256 Montgomery words go through their pre-pass first
    sm = b"".join(b32((v << 256) % ref.R) for v in ref.bytes_to_scalars(sc))
    ctx.set_scalar_format(True)
    try:
        assert ctx.msm(sm).to_affine_bytes() == want
    finally:
        ctx.set_scalar_format(False)
    ctx.set_bases(points)  # back to the plain mode on the same context
    assert not ctx.uses_endomorphism() and ctx.msm(sc).to_affine_bytes() == want


def test_input_contract_is_the_plain_mode_s(ctx):
    # non-canonical scalars (r <= s < 2^255) are plain integers in both modes; scalars that overflow the reference's recode
    # ("final carry is 1", test/utils.rs:150-152) are rejected in both; the two base flags exclude each other
    points = cpu.sample_points(94, 8)
    ctx.set_bases(points, endomorphism=True)
    ks = [ref.R, ref.R + 5, 2 * ref.R + 12345, 1 << 254, 0x7FFE << 240, 0, 1, ref.R - 1]
    sc = b"".join(b32(k) for k in ks)
    assert ctx.msm(sc).to_affine_bytes() == cpu.to_affine64(cpu.cpu_msm(points, b"".join(b32(k % ref.R) for k in ks)))
    for bad in ((1 << 256) - 1, (1 << 255) - 1):  # top digit 0x7fff + carry: rejected by the plain mode too (tests/test_gpu_msm.py)
        with pytest.raises(m.MsmHipError):
            ctx.msm(b32(bad) + sc[32:])
        ctx.set_bases(points)
        with pytest.raises(m.MsmHipError):
            ctx.msm(b32(bad) + sc[32:])
        ctx.set_bases(points, endomorphism=True)
    assert ctx.msm(sc).to_affine_bytes() is not None  # the context stays usable
    with pytest.raises(m.MsmHipError):
        ctx.set_bases(points, endomorphism=True, precompute=True)
    ctx.set_bases(points)


def test_skewed_and_cancelling_inputs(ctx):
    # all-equal scalars (two slots per window hold everything), P / -P pairs, duplicates
    n = 20000
    pts = ref.bytes_to_points(cpu.sample_points(95, 64))
    pl = [pts[i % 64] if (i // 64) % 2 == 0 else ref.neg(pts[i % 64]) for i in range(n)]
    points = ref.points_to_bytes(pl)
    base = ref.bytes_to_scalars(cpu.sample_scalars(96, 4))
    ctx.set_bases(points, endomorphism=True)
    for sc in (b32(base[0]) * n, b"".join(b32(base[i % 4]) for i in range(n)), b32(ref.glv_params()["lam"]) * n):
        assert ctx.msm(sc).to_affine_bytes() == cpu.to_affine64(cpu.cpu_msm(points, sc))
    ctx.set_bases(points)


def test_grumpkin(built):
    from oracle import cpu_grumpkin as cg
    from oracle import grumpkin_ref as gr

    g = m.MsmContext(0, curve="grumpkin")
    try:
        for n in (1, 300, 30000):
            points, sc = bytearray(cg.sample_points(97, n)), bytearray(cg.sample_scalars(98, n))
            sc[0:32] = b32(gr.glv_params()["lam"])
            points, sc = bytes(points), bytes(sc)
            g.set_bases(points, endomorphism=True)
            want = cg.to_affine64(cg.cpu_msm(points, sc))
            for bits in (0, 12, 16):
                g.set_window_bits(bits)
                assert g.msm(sc).to_affine_bytes() == want
            g.set_window_bits(0)
            eq_points, eq_scalars, _ = equivalent_problem(gr, points, sc) if n <= 300 else (None, None, None)
            if eq_points:
                assert cg.to_affine64(cg.cpu_msm(eq_points, eq_scalars)) == want
    finally:
        g.close()


def test_sharded_pipeline_over_half_length_windows(ctx):
    """ShardedMsmPipeline(halves=True): the ranks share the 8 half-length windows (msm_hip_launch_half_windows_batch_device_bn254).
    One rank doing all 8 gives whole results; the emulated share of an 8-rank run (1 window, 8 MSMs per launch) gives the partial sums
    the plain problem's stage model predicts for that window."""
    import torch

    import msm_webgpu_amd as m
    from msm_webgpu_amd.sharding import ShardedMsmPipeline, msms_per_launch

    n = 7000
    points = cpu.sample_points(95, n)
    vecs_host = [cpu.sample_scalars(96 + k, n) for k in range(8)]
    want = [cpu.to_affine64(cpu.cpu_msm(points, s)) for s in vecs_host]
    dev = torch.device("cuda", ctx.device)
    vecs = [torch.frombuffer(bytearray(s), dtype=torch.uint8).view(n, 32).to(dev) for s in vecs_host]
    ctx.set_bases(points, endomorphism=True)
    try:
        pipe = ShardedMsmPipeline(ctx, 0, 1, halves=True, msms_per_issue=8)
        assert pipe.num_windows == 8 and (pipe.w_begin, pipe.w_end) == (0, 8)
        batch = torch.cat(vecs, dim=0).contiguous()
        pipe.issue(batch, n)
        pipe.issue(batch[: 3 * n], n)
        got, got3 = pipe.complete(), pipe.complete()
        assert [g.to_affine_bytes() for g in got] == want and [g.to_affine_bytes() for g in got3] == want[:3]
        assert msms_per_launch(8, 8) == 8
        # one rank's share of an 8-rank run: half-length window 0 of 8 MSMs per launch
        pipe8 = ShardedMsmPipeline(ctx, 0, 1, halves=True, msms_per_issue=8, emulate_world=8)
        assert (pipe8.w_begin, pipe8.w_end) == (0, 1)
        pipe8.issue(batch, n)
        part = pipe8.complete()
        out = torch.empty((8, 96), dtype=torch.uint8, device=dev)
        ctx.launch_half_windows_batch(batch, n, 0, 1, 3, out)
        ctx.slot_sync(3)
        host = out.cpu()
        for k in range(8):
            assert part[k] == m.MsmContext.combine_windows(host[k:k + 1]), k
    finally:
        ctx.set_bases(points)


def test_the_drop_in_default_is_the_fast_mode_and_the_same_group_element(ctx):
    """Round 4: a base set handed over WITHOUT flags (the C ABI's flags = 0, the one-shot msm_hip_msm_bn254_g1 ≙ compute_msm, the Python
    compute_msm / run_webgpu_msm) takes the curve's endomorphism mode -- the mode the headline figure is measured in -- and
    MSM_HIP_BASES_PLAIN asks for the reference's 16-window shape.  Same result in every mode, against the oracle."""
    import ctypes as C

    n = 5000
    points, sc = cpu.sample_points(96, n), cpu.sample_scalars(97, n)
    want = cpu.to_affine64(cpu.cpu_msm(points, sc))
    ctx.set_bases(points, endomorphism=None)  # flags = 0
    assert ctx.uses_endomorphism() and ctx.msm(sc).to_affine_bytes() == want
    ctx.set_bases(points)                     # this wrapper's default: MSM_HIP_BASES_PLAIN
    assert not ctx.uses_endomorphism() and ctx.msm(sc).to_affine_bytes() == want
    ctx.set_bases(points, check_on_curve=True, endomorphism=None)  # other flags do not switch the default off
    assert ctx.uses_endomorphism() and ctx.msm(sc).to_affine_bytes() == want
    # the reference-shaped calls
    assert m.compute_msm(points, sc).to_affine_bytes() == want and m.run_webgpu_msm(points, sc).to_affine_bytes() == want
    out = C.create_string_buffer(96)
    assert m.lib().msm_hip_msm_bn254_g1(points, sc, n, out) == 0
    assert m.G1(out.raw, ref.P).to_affine_bytes() == want
    m.lib().msm_hip_oneshot_release()
    # plain and an explicit mode exclude each other; unknown flag bits are rejected
    for flags in (16 | 8, 16 | 4, 16 | 32, 64, 1 << 31):
        assert m.lib().msm_hip_set_bases_bn254(ctx._h, points, n, flags) == -2
    # fixed-base tables stay what they were; a G2 context (no endomorphism yet) takes the plain shape by default
    ctx.set_bases(points, precompute=True)
    assert not ctx.uses_endomorphism() and ctx.msm(sc).to_affine_bytes() == want
    ctx.set_bases(points)
