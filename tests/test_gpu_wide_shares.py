"""Wide fixed-base tables behind the window-sharded / multi-GPU entry points (round 5; SURVEY.md 8e + 8f-2; anchors: the reference's time / space
trade-off note README.md:70-71, its hard-coded chunk_size src/cuzk/msm.rs:79-82, its final combine src/cuzk/msm.rs:411-416).

A rank's share is a range of the VIRTUAL windows of the one bucket set (msm_hip_launch_vwindows_batch_device): at 19-bit digits 8 virtual windows
-- one per rank at 8 GPUs.  Every rank's launch runs here one after the other on the one GPU of the box; what the all-gather would deliver
((weighted sum, plain total) pairs in rank order) is finished by msm_hip_combine_vwindows_batch_curve and compared with the CPU oracle."""
import os

import numpy as np
import pytest
import torch

import msm_webgpu_amd as m
from msm_webgpu_amd.sharding import ShardedMsmPipeline, gathered_window_sums, window_range
from oracle import cpu
from tests.util import R

pytestmark = pytest.mark.gpu

THREADS = max(1, min(16, os.cpu_count() or 1))


def _host(t):
    return t.cpu().numpy().tobytes()


def _shares(ctx, batch, n, g, world):
    """all ranks' launches of `g` MSMs' shares, gathered in rank order -> [G1] * g"""
    nwin = ctx.virtual_windows()
    per = -(-nwin // world)
    gathered = torch.zeros((world, g * per * 2, ctx.jb), dtype=torch.uint8, device=batch.device)
    for rank in range(world):
        b, e = window_range(rank, world, nwin)
        if e > b:
            ctx.launch_vwindows_batch(batch, n, b, e, rank % 3, gathered[rank][: g * (e - b) * 2])
            ctx.slot_sync(rank % 3)
    pairs = gathered.cpu().numpy().reshape(world, g * per, 2 * ctx.jb)
    return m.MsmContext.combine_vwindows_batch(gathered_window_sums(pairs, g, world, nwin), nwin, ctx.curve)


@pytest.fixture(scope="module")
def inputs_2p20(ctx):
    n = 1 << 20
    pts = ctx.sample_points(n, 0xD2_0001)
    sets = [ctx.sample_scalars(n, 0xD2_0100 + k) for k in range(2)]
    pb = _host(pts)
    want = [cpu.to_affine64(cpu.cpu_msm(pb, _host(s), THREADS)) for s in sets]
    yield n, pts, sets, want
    ctx.set_wide_bits(0)


@pytest.mark.parametrize("bits,world,g", [(19, 8, 8), (19, 4, 4), (19, 3, 2), (20, 8, 4), (17, 2, 2)])
def test_2p20_virtual_window_shares_match_oracle(ctx, inputs_2p20, bits, world, g):
    n, pts, sets, want = inputs_2p20
    ctx.set_wide_bits(bits)
    ctx.set_bases(pts, precompute="wide")
    assert ctx.wide_bits() == bits and ctx.virtual_windows() == 1 << (bits - 16)
    batch = torch.cat([sets[k & 1] for k in range(g)], dim=0).contiguous()
    got = _shares(ctx, batch, n, g, world)
    assert [r.to_affine_bytes() for r in got] == [want[v & 1] for v in range(g)]
    # whole MSMs on the same tables still agree (the share launches leave nothing behind in the context)
    assert ctx.msm(sets[1]).to_affine_bytes() == want[1]


def test_2p20_skewed_scalars_through_shares(ctx, inputs_2p20):
    """every scalar equal (all entries of a digit position in ONE virtual window: the other ranks' shares are empty), a witness-like vector,
    and the largest scalars of the field"""
    n, pts, sets, _ = inputs_2p20
    pb = _host(pts)
    s = 0x123456789ABCDEF013579BDF2468ACE0FEDCBA9876543210
    eq = torch.tensor(list(s.to_bytes(32, "little")), dtype=torch.uint8, device=pts.device).repeat(n, 1).contiguous()
    gen = torch.Generator(device=pts.device)
    gen.manual_seed(11)
    sel = torch.rand(n, device=pts.device, generator=gen)
    wit = sets[0].clone()
    wit[sel < 0.7] = 0
    wit[(sel >= 0.4) & (sel < 0.7), 0] = 1
    top = torch.tensor(list((R - 1).to_bytes(32, "little")), dtype=torch.uint8, device=pts.device).repeat(n, 1).contiguous()
    top[::2] = torch.tensor(list((R - 2 ** 200 - 7).to_bytes(32, "little")), dtype=torch.uint8, device=pts.device)
    ctx.set_wide_bits(19)
    ctx.set_bases(pts, precompute="wide")
    for name, sc in (("all equal", eq), ("witness-like", wit), ("largest", top)):
        want = cpu.to_affine64(cpu.cpu_msm(pb, _host(sc), THREADS))
        # 2 ranks: four virtual windows per share -- with equal scalars several digit positions' 2048 entries per block iteration land in one
        # share, more than the scatter's LDS staging holds at once (WIDE_SHARE_CAP: the staging-window loop)
        for world in (8, 2):
            assert _shares(ctx, sc, n, 1, world)[0].to_affine_bytes() == want, (name, world)


@pytest.mark.parametrize("bits", [16, 17, 18, 19, 20])
@pytest.mark.parametrize("n", [1, 3, 300, 70000])
def test_small_and_odd_sizes_every_width(ctx, bits, n):
    pts, sc = ctx.sample_points(n, 0xD3_0000 + n), ctx.sample_scalars(3 * n, 0xD3_1000 + n)
    pb = _host(pts)
    want = [cpu.to_affine64(cpu.cpu_msm(pb, _host(sc[k * n:(k + 1) * n]), 4)) for k in range(3)]
    ctx.set_wide_bits(bits)
    try:
        ctx.set_bases(pts, precompute="wide")
        nwin = ctx.virtual_windows()
        for world in sorted({1, 2, min(8, nwin), nwin}):
            got = _shares(ctx, sc, n, 3, world)
            assert [r.to_affine_bytes() for r in got] == want, (bits, n, world)
    finally:
        ctx.set_wide_bits(0)


def test_edge_scalars_and_errors(ctx):
    n = 64
    pts = ctx.sample_points(n, 0xD4_0001)
    pb = _host(pts)
    vals = [0, 1, R - 1, R - 2, 1 << 253, (1 << 253) - 1, 0x8000, 0x7fff, 1 << 18, (1 << 18) + 1, (1 << 19) - 1, 0x40000 << 19, 0x3ffff << 38]
    sb = b"".join(int(vals[i % len(vals)] if i % 3 else (vals[i % len(vals)] * 0x10001 + i) % R).to_bytes(32, "little") for i in range(n))
    sc = torch.tensor(list(sb), dtype=torch.uint8, device=pts.device).view(n, 32)
    want = cpu.to_affine64(cpu.cpu_msm(pb, sb, 2))
    for bits in (17, 19, 20):
        ctx.set_wide_bits(bits)
        ctx.set_bases(pts, precompute="wide")
        assert _shares(ctx, sc, n, 1, 8 if bits > 17 else 2)[0].to_affine_bytes() == want, bits
    out = torch.zeros((64, ctx.jb), dtype=torch.uint8, device=pts.device)
    nwin = ctx.virtual_windows()
    with pytest.raises(m.MsmHipError):  # range beyond the virtual windows
        ctx.launch_vwindows_batch(sc, n, 0, nwin + 1, 0, out)
    with pytest.raises(m.MsmHipError):  # empty range
        ctx.launch_vwindows_batch(sc, n, 3, 3, 0, out)
    ctx.launch_vwindows_batch(sc, n, 0, 1, 0, out)
    with pytest.raises(m.MsmHipError):  # a share is not a whole MSM: finish refuses it
        ctx.finish(0)
    ctx.slot_sync(0)
    # a scalar at or above the modulus is rejected, as on every other path
    bad = sc.clone()
    bad[5] = torch.tensor(list(R.to_bytes(32, "little")), dtype=torch.uint8, device=pts.device)
    bad[5, 31] = 0xFF
    ctx.launch_vwindows_batch(bad, n, 0, nwin, 1, out)
    with pytest.raises(m.MsmHipError):
        ctx.slot_sync(1)
    ctx.set_wide_bits(0)
    ctx.set_bases(pts)
    with pytest.raises(m.MsmHipError):  # bases without wide tables
        ctx.launch_vwindows_batch(sc, n, 0, 1, 0, out)


def test_pipeline_with_wide_shares_single_rank(ctx):
    """ShardedMsmPipeline(wide=True) with one rank: all virtual windows, pairs through the pinned buffer, the library's finish"""
    n = 5000
    pts, sc = ctx.sample_points(n, 0xD5_0001), ctx.sample_scalars(4 * n, 0xD5_0002)
    pb = _host(pts)
    want = [cpu.to_affine64(cpu.cpu_msm(pb, _host(sc[k * n:(k + 1) * n]), 4)) for k in range(4)]
    ctx.set_wide_bits(19)
    try:
        ctx.set_bases(pts, precompute="wide")
        pipe = ShardedMsmPipeline(ctx, 0, 1, depth=2, msms_per_issue=4, wide=True)
        assert pipe.num_windows == 8 and pipe.rec == 2
        pipe.issue(sc, n)
        pipe.issue(sc[: 2 * n].contiguous(), n)
        assert [r.to_affine_bytes() for r in pipe.complete()] == want
        assert [r.to_affine_bytes() for r in pipe.complete()] == want[:2]
    finally:
        ctx.set_wide_bits(0)


@pytest.mark.parametrize("bits", [0, 20])
def test_2p18_through_the_multi_gpu_abi(bits):
    """msm_hip_mgpu_* with wide tables (8 contexts on the one GPU, pinned-buffer gather): the shares are virtual windows, 8 MSMs per launch"""
    n = 1 << 18
    c0 = m.MsmContext(0)
    pts = c0.sample_points(n, 0xD6_0001)
    sets = [c0.sample_scalars(n, 0xD6_0100 + k) for k in range(2)]
    c0.close()
    pb = _host(pts)
    want = [cpu.to_affine64(cpu.cpu_msm(pb, _host(s), THREADS)) for s in sets]
    mg = m.MultiGpuMsm([0] * 8, "host")
    try:
        mg.set_wide_bits(bits)
        mg.set_bases(pb, precompute="wide")
        g = mg.group_size
        assert g == 8
        batch = torch.cat([sets[k & 1] for k in range(g)], dim=0).contiguous()
        for k in range(3):
            assert mg.launch_batch([batch] * 8, n, k) == g
        for k in range(3):
            assert [r.to_affine_bytes() for r in mg.finish_batch(k, g)] == [want[v & 1] for v in range(g)], k
        hb = _host(batch[: 3 * n])
        assert mg.launch_batch(hb, n, 1) == 3
        assert [r.to_affine_bytes() for r in mg.finish_batch(1, 3)] == [want[0], want[1], want[0]]
        assert mg.msm(_host(sets[1])).to_affine_bytes() == want[1]
        # whole MSMs dealt out over the devices run on the same tables
        assert [r.to_affine_bytes() for r in mg.msm_batch(_host(batch[: 2 * n]), n)] == want
    finally:
        mg.close()


def test_bls12_381_shares():
    """another curve through the same entry points (144-byte records; 19-bit digits are the narrowest its scalar field admits)"""
    from oracle import cpu_bls12_381 as c

    n = 3000
    ctx = m.MsmContext(0, "bls12_381")
    try:
        pb, sb = c.sample_points(61, n), c.sample_scalars(62, 2 * n)
        sc = torch.tensor(list(sb), dtype=torch.uint8, device="cuda:0").view(2 * n, 32)
        want = [c.to_affine64(c.cpu_msm(pb, sb[k * 32 * n:(k + 1) * 32 * n], 4)) for k in range(2)]
        ctx.set_wide_bits(19)
        ctx.set_bases(pb, precompute="wide")
        got = _shares(ctx, sc, n, 2, 8)
        assert [r.to_affine_bytes() for r in got] == want
    finally:
        ctx.close()
