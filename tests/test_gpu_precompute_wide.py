"""Wide fixed-base tables (round 4): MSM_HIP_BASES_PRECOMPUTE_WIDE stores 2^(C w) P_i and recodes every scalar into signed digits
of C = 17 bits (15 of them; 16 bits up to 2^16 points, 20 bits -- 13 digits -- beyond 2^20) -- 15 (16, 13) bucket additions per point instead of the reference's 16
(src/cuzk/msm.rs:79-82) -- into one bucket set of 2^(C-1) slots that the engine runs as 2 (8) virtual windows of 2^15.  Same group element as
the plain engine and the oracle, at every digit width the library builds (17 .. 20)."""
import pytest

import msm_webgpu_amd as m
from oracle import bn254_ref as ref
from oracle import cpu
from tests.util import R, case_inputs, golden_cases

pytestmark = pytest.mark.gpu

WIDTHS = [0, 16, 17, 18, 19, 20]  # 0: the width the library picks from the number of bases


@pytest.fixture(params=WIDTHS)
def wctx(ctx, request):
    ctx.set_wide_bits(request.param)
    yield ctx
    ctx.set_wide_bits(0)


def test_golden_vectors_with_wide_tables(wctx):
    ctx = wctx
    for case in golden_cases():
        points, scalars = case_inputs(case)
        ctx.set_bases(points, check_on_curve=True, precompute="wide")
        assert ctx.msm(scalars).to_affine_bytes().hex() == case["expected_affine"], case["name"]


def test_width_follows_the_base_count(ctx):
    pts = ctx.sample_points(300, 1399)
    ctx.set_bases(pts, precompute="wide")
    assert ctx.wide_bits() == 16
    big = ctx.sample_points((1 << 16) + 1, 1398)
    ctx.set_bases(big, precompute="wide")
    assert ctx.wide_bits() == 17
    ctx.set_wide_bits(19)
    ctx.set_bases(pts, precompute="wide")
    assert ctx.wide_bits() == 19
    ctx.set_wide_bits(0)
    ctx.set_bases(pts)
    assert ctx.wide_bits() == 0
    for bad in (15, 21, -1):
        assert m.lib().msm_hip_set_wide_bits(ctx._h, bad) == -2


@pytest.mark.parametrize("n", [1, 2, 257, 5000, 1 << 16, (1 << 18) + 3])
def test_wide_tables_match_oracle_and_plain_engine(wctx, n):
    ctx = wctx
    if n > 5000 and ctx.wide_bits_choice not in (0, 16, 19):
        pytest.skip("large sizes at the two widths the policy uses")
    pts, sc = ctx.sample_points(n, 1400 + n), ctx.sample_scalars(n, 1401 + n)
    pb, sb = pts.cpu().numpy().tobytes(), sc.cpu().numpy().tobytes()
    want = cpu.to_affine64(cpu.cpu_msm(pb, sb, 8))
    ctx.set_bases(pts, precompute="wide")
    assert ctx.msm(sc).to_affine_bytes() == want          # device scalars
    assert ctx.msm(sb).to_affine_bytes() == want          # host scalars
    k = max(1, n // 3)                                    # a prefix of the bases: table stride stays n
    assert ctx.msm(sc[:k].contiguous()).to_affine_bytes() == cpu.to_affine64(cpu.cpu_msm(pb[: 64 * k], sb[: 32 * k], 8))
    # the window-sharding entry points ignore the tables (table 0 is the plain base set)
    parts = [ctx.msm_windows(sc, 0, 7), ctx.msm_windows(sc, 7, 16)]
    import torch
    assert m.MsmContext.combine_windows(torch.cat(parts, dim=0)).to_affine_bytes() == want
    ctx.set_bases(pts)                                     # back to plain: the flag does not stick
    assert ctx.msm(sc).to_affine_bytes() == want


def test_wide_digit_edges(wctx):
    ctx = wctx
    """digits at the seams of the virtual windows (magnitude k * 2^15, k * 2^15 +- 1), the most negative digit (with its carry), in every
    window position (17- to 20-bit grids), beside the largest scalars the input contract admits"""
    vals = []
    for bits, count in ((16, 16), (17, 15), (18, 15), (19, 14), (20, 13)):
        for w in range(count):
            for d in (1, 0x7fff, 0x8000, 0x8001, 0xffff, 0x10000, 0x10001, 0x3ffff, 0x40000, 0x40001, 0x78000, 0x78001, 0x7ffff, 0x80000, 0xf8001, 0xfffff):
                v = d << (bits * w)
                if v < R:
                    vals.append(v)
    vals += [R - 1, R - 2, 0, (1 << 253) - 1, int("8000" * 15, 16), int("7ffff" * 12, 16), int("80000" * 12, 16), int("fffff" * 12, 16),
             sum(0x3ffff << (19 * w) for w in range(13)), sum(0x40000 << (19 * w) for w in range(13)), sum(0x7ffff << (19 * w) for w in range(13))]
    n = len(vals)
    points = cpu.sample_points(1410, n)
    sb = ref.scalars_to_bytes(vals)
    want = cpu.to_affine64(cpu.cpu_msm(points, sb, 8))
    ctx.set_bases(points, precompute="wide")
    assert ctx.msm(sb).to_affine_bytes() == want
    # each of them alone (a one-point MSM names the failing digit)
    for i in range(0, n, 23):
        one = cpu.to_affine64(cpu.cpu_msm(points[: 64 * (i + 1)], bytes(32 * i) + sb[32 * i: 32 * i + 32], 8))
        assert ctx.msm(bytes(32 * i) + sb[32 * i: 32 * i + 32]).to_affine_bytes() == one, hex(vals[i])


def test_wide_tables_extreme_and_skewed_scalars(wctx):
    ctx = wctx
    n = 20000
    points = cpu.sample_points(1411, n)
    s = 0x0FED_CBA9_8765_4321_0F1E_2D3C_4B5A_6978_8796_A5B4_C3D2_E1F0 % R
    ctx.set_bases(points, precompute="wide")
    # one giant bucket per window + edge digits; witness-like (zeros, ones, small values: everything in virtual window 0); duplicates of one point's worth
    for vals in ([s] * (n - 8) + [R - 1, 0, 1, 0x8000, (1 << 253) - 1, int("8000" * 15, 16), 2, R - 2],
                 [(i * 7919) % 3 for i in range(n)],
                 [(i * 104729) % 70000 for i in range(n)]):
        sb = ref.scalars_to_bytes(vals)
        assert ctx.msm(sb).to_affine_bytes() == cpu.to_affine64(cpu.cpu_msm(points, sb, 8))
    with pytest.raises(m.MsmHipError) as e:
        ctx.msm(b"\xff" * 32)
    assert e.value.code == -4
    # a top digit that does not fit after its shift (far above the modulus, at every digit width): rejected, not mis-added; up to 2^254 + a
    # little the shifted digit still fits
    p0 = ref.bytes_to_points(points[:64])[0]
    for v in ((1 << 254) - 1, (1 << 254) + 5):   # (at 17 bits the top digit is exactly 2^16 here: the largest bucket magnitude, beyond the 17-bit field)
        # (beyond the modulus, where the C oracle's Booth windows -- halo2curves' -- end: the big-integer model is the reference here)
        assert ctx.msm(v.to_bytes(32, "little")).to_affine_bytes() == ref.affine_to_bytes64(ref.mul(v % R, p0)), hex(v)
    # further above the modulus: rejected (the top digit does not fit the bucket set) or still summed exactly -- never mis-added
    for v in ((1 << 254) + (1 << 253), (1 << 255) + (1 << 254)):
        try:
            got = ctx.msm(v.to_bytes(32, "little"))
        except m.MsmHipError as e:
            assert e.code == -4, hex(v)
        else:
            assert got.to_affine_bytes() == ref.affine_to_bytes64(ref.mul(v % R, p0)), hex(v)


def test_wide_tables_batches_and_flags(wctx):
    ctx = wctx
    n, batch = 3000, 14
    pts = ctx.sample_points(n, 1420)
    sc = ctx.sample_scalars(n * batch, 1421)
    ctx.set_bases(pts, precompute="wide")
    # an MSM is 2^(C-16) local windows, and a launch leaves bit-plane sums for at most 24: 24 / 12 / 6 / 3 / 1 whole MSMs per launch at 16 .. 20 bits
    group = 24 >> (ctx.wide_bits() - 16)
    assert ctx.batch_group_size(n) == group
    got = ctx.msm_batch(sc, n)                             # groups of `group`, the last one shorter
    pb = pts.cpu().numpy().tobytes()
    for k in (0, 2, 4, 11, 12, 13):
        assert got[k].to_affine_bytes() == cpu.to_affine64(cpu.cpu_msm(pb, sc[k * n:(k + 1) * n].cpu().numpy().tobytes())), k
    assert [g.to_affine_bytes() for g in ctx.msm_batch(sc.cpu().numpy().tobytes(), n)] == [g.to_affine_bytes() for g in got]
    g2 = min(group, 5)
    assert ctx.launch_batch(sc[: g2 * n].contiguous(), n, 2) == g2
    assert [x.to_affine_bytes() for x in ctx.finish_batch(2, g2)] == [x.to_affine_bytes() for x in got[:g2]]
    if group < 14:   # more whole MSMs than one launch's plane sums hold: refused, the slot stays free
        with pytest.raises(m.MsmHipError):
            ctx.launch_batch(sc[: (group + 1) * n].contiguous(), n, 3)
    # pipelined launches through the result slots
    for k in range(4):
        ctx.launch(sc[k * n:(k + 1) * n].contiguous(), slot=k)
    assert [ctx.finish(k).to_affine_bytes() for k in range(4)] == [g.to_affine_bytes() for g in got[:4]]
    # scalars handed over as s * 2^256 mod r (the conversion pass sits in front of the wide recode as in front of every other)
    vals = ref.bytes_to_scalars(sc[:n].cpu().numpy().tobytes())
    ctx.set_scalar_format(True)
    try:
        assert ctx.msm(ref.scalars_to_bytes([(v << 256) % R for v in vals])).to_affine_bytes() == got[0].to_affine_bytes()
    finally:
        ctx.set_scalar_format(False)
    # not combinable with the other table mode or the endomorphism
    for flags in (32 | 4, 32 | 8, 32 | 16):
        assert m.lib().msm_hip_set_bases_bn254(ctx._h, pb, n, flags) == -2
    ctx.set_bases(pts)
    assert ctx.msm(sc[:n].contiguous()).to_affine_bytes() == got[0].to_affine_bytes()


@pytest.mark.parametrize("n,bits", [(1 << 20, 17), ((1 << 20) + 8, 20)])
def test_wide_tables_at_full_sizes_match_the_oracle(ctx, n, bits):
    """BASELINE config 2's size (and just beyond it, where the policy gives 20-bit digits): the wide tables' result against the CPU ORACLE on the
    same inputs (≙ src/lib.rs:166: GPU == cpu_msm), for uniform and for skewed scalars -- and the endomorphism mode's and the 16-bit tables'
    beside it: three independent paths through the sort and the finish."""
    import os

    import torch

    threads = max(1, min(16, os.cpu_count() or 1))
    pts = ctx.sample_points(n, 1440)
    uniform = ctx.sample_scalars(n, 1441)
    skew = uniform.clone()
    skew[: n // 2, 2:] = 0                       # half of the scalars below 2^16: the lowest digit position takes far more than its share
    skew[n // 2: n // 2 + n // 8] = uniform[7]   # and an eighth of them equal: giant buckets in every digit position
    pb = pts.cpu().numpy().tobytes()
    want = {name: cpu.to_affine64(cpu.cpu_msm(pb, sc.cpu().numpy().tobytes(), threads)) for name, sc in (("uniform", uniform), ("skew", skew))}
    for mode in ("wide", "endomorphism", "tables"):
        ctx.set_bases(pts, endomorphism=mode == "endomorphism", precompute="wide" if mode == "wide" else mode == "tables")
        if mode == "wide":
            assert ctx.wide_bits() == bits
        for name, sc in (("uniform", uniform), ("skew", skew)):
            assert ctx.msm(sc).to_affine_bytes() == want[name], (mode, name)
    del skew, uniform, pts
    torch.cuda.empty_cache()
    ctx.set_bases(ctx.sample_points(16, 1))      # give the tables' memory back


def test_wide_tables_2p22_policy_width(ctx):
    """2^22 points at the width the policy picks there (20 bits: 13 tables, 3.25 GiB): the whole MSM against the endomorphism mode, a 2^16 slice
    of the same inputs against the CPU oracle (on the slice's own 16-bit tables AND forced to 20 bits), and split consistency over the tables"""
    import os

    import torch

    n = 1 << 22
    threads = max(1, min(16, os.cpu_count() or 1))
    pts, sc = ctx.sample_points(n, 1450), ctx.sample_scalars(n, 1451)
    ctx.set_bases(pts, precompute="wide")
    assert ctx.wide_bits() == 20
    whole = ctx.msm(sc)
    h = n // 2 + 4321  # MSM(P, s) = MSM(P[:h], s[:h]) + the rest: the prefix runs over the same tables (table stride stays n)
    first = ctx.msm(sc[:h].contiguous())
    ctx.set_bases(pts, endomorphism=True)
    assert ctx.msm(sc) == whole
    ctx.set_bases(pts[h:].contiguous(), endomorphism=True)
    second = ctx.msm(sc[h:].contiguous())
    assert ref.add(first.to_affine(), second.to_affine()) == whole.to_affine()
    k, off = 1 << 16, 3 << 20
    sl_p, sl_s = pts[off:off + k].contiguous(), sc[off:off + k].contiguous()
    want = cpu.to_affine64(cpu.cpu_msm(sl_p.cpu().numpy().tobytes(), sl_s.cpu().numpy().tobytes(), threads))
    for bits in (0, 20):
        ctx.set_wide_bits(bits)
        ctx.set_bases(sl_p, precompute="wide")
        assert ctx.msm(sl_s).to_affine_bytes() == want, bits
    ctx.set_wide_bits(0)
    del pts, sc
    torch.cuda.empty_cache()
    ctx.set_bases(ctx.sample_points(16, 1))


def test_wide_tables_behind_the_multi_gpu_abi(built):
    """BASELINE config 5's shape through msm_hip_mgpu_run_batch_bn254: whole MSMs over one fixed base dealt out over the devices (here three
    contexts on one GPU), every device holding the wide tables; the window-sharded single MSM on the same handle ignores them."""
    n, batch = 4000, 7
    points = cpu.sample_points(1430, n)
    sc = cpu.sample_scalars(1431, n * batch)
    mg = m.MultiGpuMsm([0, 0, 0], "host")
    try:
        mg.set_bases(points, precompute="wide")
        got = mg.msm_batch(sc, n)
        for k in (0, 3, 6):
            assert got[k].to_affine_bytes() == cpu.to_affine64(cpu.cpu_msm(points, sc[32 * n * k: 32 * n * (k + 1)], 8)), k
        assert mg.msm(sc[: 32 * n]).to_affine_bytes() == got[0].to_affine_bytes()
    finally:
        mg.close()
