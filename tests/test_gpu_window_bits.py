"""Adaptive window size (the reference hard-codes c, src/cuzk/msm.rs:79-82): the same group element for every
window size, bit-exact against the oracle -- golden vectors, edge sizes, batches -- and the automatic choice by n."""
import pytest
import torch

import msm_webgpu_amd as m
from oracle import bn254_ref as ref
from oracle import cpu
from tests.util import R, case_inputs, golden_cases

pytestmark = pytest.mark.gpu


@pytest.fixture(params=[12, 14, 16])
def bits(ctx, request):
    ctx.set_window_bits(request.param)
    yield request.param
    ctx.set_window_bits(0)


def test_window_shapes(ctx):
    assert [ctx.window_config(b) for b in (12, 14, 16)] == [(22, 2048), (19, 8192), (16, 32768)]
    with pytest.raises(m.MsmHipError):
        ctx.set_window_bits(13)


def test_golden_vectors_every_window_size(ctx, bits):
    for case in golden_cases():
        points, scalars = case_inputs(case)
        ctx.set_bases(points)
        got = ctx.msm(scalars)
        assert ctx.last_window_bits() == bits or not scalars
        assert got.to_affine_bytes().hex() == case["expected_affine"], (bits, case["name"])


@pytest.mark.parametrize("n", [1, 63, 257, 5000, 40000, 1 << 17])
def test_sizes_every_window_size(ctx, bits, n):
    pts, sc = ctx.sample_points(n, 300 + n), ctx.sample_scalars(n, 301 + n)
    ctx.set_bases(pts)
    want = cpu.to_affine64(cpu.cpu_msm(pts.cpu().numpy().tobytes(), sc.cpu().numpy().tobytes(), 8))
    assert ctx.msm(sc).to_affine_bytes() == want
    assert ctx.msm(sc.cpu().numpy().tobytes()).to_affine_bytes() == want  # host-scalar entry point


def test_extreme_digits_every_window_size(ctx, bits):
    # r - 1, 2^253, all-ones digits and the digit -2^(c-1) chain for this window size
    W, H = ctx.window_config(bits)
    chain = sum(H << (bits * w) for w in range(W)) % (1 << 253)
    vals = [R - 1, R - 2, 1 << 253, (1 << 253) - 1, chain, H, H - 1, H + 1, (1 << bits) - 1, 1 << bits, 0, 1]
    points = cpu.sample_points(310, len(vals))
    sb = ref.scalars_to_bytes(vals)
    ctx.set_bases(points)
    assert ctx.msm(sb).to_affine_bytes() == cpu.to_affine64(cpu.cpu_msm(points, sb))
    with pytest.raises(m.MsmHipError) as e:  # 2^256 - 1 does not fit any recode
        ctx.msm(b"\xff" * 32)
    assert e.value.code == -4


def test_skew_every_window_size(ctx, bits):
    n = 30000
    points = cpu.sample_points(320, n)
    s = 0x0123_4567_89AB_CDEF_0F1E_2D3C_4B5A_6978_8796_A5B4_C3D2_E1F0 % R
    sb = s.to_bytes(32, "little") * n  # every entry of every window in one bucket
    ctx.set_bases(points)
    assert ctx.msm(sb).to_affine_bytes() == cpu.to_affine64(cpu.cpu_msm(points, sb, 8))


def test_batches_every_window_size(ctx, bits):
    n, batch = 3000, 9
    pts = ctx.sample_points(n, 330)
    sc = ctx.sample_scalars(n * batch, 331)
    ctx.set_bases(pts)
    got = ctx.msm_batch(sc, n)
    pb = pts.cpu().numpy().tobytes()
    for k in (0, 4, 8):
        assert got[k].to_affine_bytes() == cpu.to_affine64(cpu.cpu_msm(pb, sc[k * n:(k + 1) * n].cpu().numpy().tobytes())), k
    assert [g.to_affine_bytes() for g in ctx.msm_batch(sc.cpu().numpy().tobytes(), n)] == [g.to_affine_bytes() for g in got]
    # as many whole MSMs per launch as the window size allows
    g = ctx.batch_group_size(n)
    assert g == 64 // ctx.window_config(bits)[0]
    assert ctx.launch_batch(sc[: g * n].contiguous(), n, 1) == g
    assert [x.to_affine_bytes() for x in ctx.finish_batch(1, g)] == [x.to_affine_bytes() for x in got[:g]]


def test_automatic_choice_follows_n(ctx):
    ctx.set_window_bits(0)
    pts = ctx.sample_points(1 << 18, 340)
    sc = ctx.sample_scalars(1 << 18, 341)
    ctx.set_bases(pts)
    seen = {}
    for logn in (10, 12, 13, 16, 18):
        ctx.msm(sc[: 1 << logn].contiguous())
        seen[logn] = ctx.last_window_bits()
    assert seen == {10: 12, 12: 12, 13: 16, 16: 16, 18: 16}  # measured optimum: profiles/r02_window_bits_latency.txt
    # the window-sharding entry points keep the reference's 16-bit windows whatever n is
    ctx.msm_windows(sc[:1000].contiguous(), 3, 5)
    assert ctx.last_window_bits() == 16
    # several whole MSMs per launch: 14 bits up to 2^16 points (profiles/r02_window_bits_grouped.txt), as many MSMs as fit 64 local windows
    assert ctx.batch_group_size(2000) == 3 and ctx.batch_group_size(1 << 17) == 4
    three = torch.cat([sc[:2000]] * 3, dim=0).contiguous()
    ctx.launch_batch(three, 2000, 0)
    assert ctx.last_window_bits() == 14
    r3 = ctx.finish_batch(0, 3)
    # 4 whole MSMs in one launch only fit 16-bit windows: the launch falls back to them
    small = torch.cat([sc[:2000]] * 4, dim=0).contiguous()
    ctx.launch_batch(small, 2000, 0)
    assert ctx.last_window_bits() == 16
    r = ctx.finish_batch(0, 4)
    assert r[0] == r[3] == r3[0] == r3[2] == ctx.msm(sc[:2000].contiguous())
    # the same with the endomorphism's half-length scalars: 10 windows of 14 bits, 6 MSMs per launch
    ctx.set_bases(pts, endomorphism=True)
    assert ctx.batch_group_size(2000) == 6 and ctx.batch_group_size(1 << 17) == 8
    six = torch.cat([sc[:2000]] * 6, dim=0).contiguous()
    ctx.launch_batch(six, 2000, 1)
    assert ctx.last_window_bits() == 14
    assert ctx.finish_batch(1, 6)[5] == r[0]
    ctx.set_bases(pts)
