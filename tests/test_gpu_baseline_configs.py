"""BASELINE.json's configs at their REAL sizes, through the C ABI on one MI355X (≙ the reference's own end-to-end test at
its largest size, tests/test_webgpu_msm_cuzk_20.rs:9-12 -> src/lib.rs:152-167: GPU MSM == cpu_msm).

  C2  2^20 points, 1 GPU                      : bit-exact against the multithreaded CPU oracle
  C3  2^20 points, windows over 8 GPUs + gather: all 8 ranks' shares (what each rank launches: 8 MSMs x 2 windows per
                                                launch), gathered in rank order and combined == whole MSM == oracle
  C4  2^24 points, 1 GPU                      : split consistency, the 8 window ranges combined, oracle on a 2^16 slice
  C5  64 x 2^18 over one shared base          : whole batch; 3 sampled vectors against the oracle; the 8-rank partition
The multi-GPU shapes run their per-rank launches one after the other on this single GPU: the arithmetic, the partitioning
and the reassembly are exactly the ones the ranks use; only the transport (RCCL) is absent.
"""
import os

import pytest
import torch

import msm_webgpu_amd as m
from msm_webgpu_amd.sharding import batch_range, group_window_rows, msms_per_launch, window_range
from oracle import bn254_ref as ref
from oracle import cpu

pytestmark = pytest.mark.gpu

THREADS = max(1, min(16, os.cpu_count() or 1))  # the GPU box's CPU share for one GPU


def _host(t):
    return t.cpu().numpy().tobytes()


@pytest.fixture(scope="module")
def inputs_2p20(ctx):
    n = 1 << 20
    pts = ctx.sample_points(n, 0xC2_0001)
    sets = [ctx.sample_scalars(n, 0xC2_0100 + k) for k in range(2)]
    pb = _host(pts)
    want = [cpu.to_affine64(cpu.cpu_msm(pb, _host(s), THREADS)) for s in sets]
    return n, pts, sets, want


def test_config2_2p20_single_gpu_matches_oracle(ctx, inputs_2p20):
    n, pts, sets, want = inputs_2p20
    ctx.set_bases(pts)
    for s, w in zip(sets, want):
        assert ctx.msm(s).to_affine_bytes() == w
    # the host-bytes entry point (what the Rust shim calls) and the asynchronous host launch over two slots
    hb = [_host(s) for s in sets]
    assert ctx.msm(hb[0]).to_affine_bytes() == want[0]
    ctx.launch_host(hb[0], 0)
    ctx.launch_host(hb[1], 1)
    assert ctx.finish(0).to_affine_bytes() == want[0]
    ctx.launch_host(hb[0], 0)
    assert ctx.finish(1).to_affine_bytes() == want[1]
    assert ctx.finish(0).to_affine_bytes() == want[0]


def test_config3_2p20_window_shares_of_8_ranks(ctx, inputs_2p20):
    n, pts, sets, want = inputs_2p20
    world = 8
    g = msms_per_launch(world)  # 8 MSMs' shares per launch
    per = 16 // world
    ctx.set_bases(pts)
    batch = torch.cat([sets[k & 1] for k in range(g)], dim=0).contiguous()
    gathered = torch.zeros((world, g * per, 96), dtype=torch.uint8, device=batch.device)  # what the all-gather would deliver
    for rank in range(world):
        b, e = window_range(rank, world)
        assert e - b == per
        ctx.launch_windows_batch(batch, n, b, e, rank % 3, gathered[rank])
        ctx.slot_sync(rank % 3)
    host = gathered.cpu()
    for v in range(g):
        got = m.MsmContext.combine_windows(group_window_rows(host, v, world))
        assert got.to_affine_bytes() == want[v & 1], v
    # uneven partitions (3 and 5 ranks) of one MSM
    for world in (3, 5):
        parts = [ctx.msm_windows(sets[0], *window_range(r, world)) for r in range(world)]
        assert m.MsmContext.combine_windows(torch.cat(parts, dim=0)).to_affine_bytes() == want[0], world


def test_config3_2p20_half_window_shares_with_endomorphism_bases(ctx, inputs_2p20):
    """C3 with MSM_HIP_BASES_ENDOMORPHISM: the ranks share the 8 half-length windows of the 2n-point problem (one per rank at 8 GPUs,
    8 MSMs per launch; reference shape: 16 full-length windows, src/cuzk/msm.rs:79-82) -- gathered in rank order, 8 sums per MSM."""
    from msm_webgpu_amd.sharding import gathered_window_sums

    n, pts, sets, want = inputs_2p20
    ctx.set_bases(pts, endomorphism=True)
    try:
        for world, g in ((8, 8), (4, 4), (3, 2)):
            per = -(-8 // world)
            batch = torch.cat([sets[k & 1] for k in range(g)], dim=0).contiguous()
            gathered = torch.zeros((world, g * per, 96), dtype=torch.uint8, device=batch.device)
            for rank in range(world):
                b, e = window_range(rank, world, 8)
                ctx.launch_half_windows_batch(batch, n, b, e, rank % 3, gathered[rank][: g * (e - b)])
                ctx.slot_sync(rank % 3)
            got = m.MsmContext.combine_windows_batch(gathered_window_sums(gathered.cpu().numpy(), g, world, 8), 8)
            assert [r.to_affine_bytes() for r in got] == [want[v & 1] for v in range(g)], world
        # the plain 16-window shares keep working on the same bases (records 0 .. n-1 are the plain set)
        parts = [ctx.msm_windows(sets[0], *window_range(r, 8)) for r in range(8)]
        assert m.MsmContext.combine_windows(torch.cat(parts, dim=0)).to_affine_bytes() == want[0]
        with pytest.raises(m.MsmHipError):  # half-window range out of [0, 8]
            ctx.launch_half_windows_batch(sets[0], n, 4, 9, 0, gathered[0])
    finally:
        ctx.set_bases(pts)
    with pytest.raises(m.MsmHipError):  # bases without their endomorphism images
        ctx.launch_half_windows_batch(sets[0], n, 0, 1, 0, gathered[0])


@pytest.mark.parametrize("endo", [False, True], ids=["plain", "endomorphism"])
def test_config3_2p20_through_the_multi_gpu_abi(inputs_2p20, endo):
    """C3 through msm_hip_mgpu_launch_batch_* / finish_batch (8 contexts on the one GPU, pinned-buffer gather): 8 MSMs' window shares
    per launch, several launches in flight, device-resident and host scalars; the Rust caller's shape (src/lib.rs:76-82)."""
    n, pts, sets, want = inputs_2p20
    mg = m.MultiGpuMsm([0] * 8, "host")
    try:
        mg.set_bases(_host(pts), endomorphism=endo)
        g = mg.group_size
        assert g == 8
        batch = torch.cat([sets[k & 1] for k in range(g)], dim=0).contiguous()
        per_device = [batch] * 8  # one pointer per device (here: the same GPU)
        for k in range(3):  # three launches in flight over three result slots
            assert mg.launch_batch(per_device, n, k) == g
        for k in range(3):
            got = mg.finish_batch(k, g)
            assert [r.to_affine_bytes() for r in got] == [want[v & 1] for v in range(g)], k
        # host scalars, fewer vectors than a full group, a slot reused
        hb = _host(batch[: 3 * n])
        assert mg.launch_batch(hb, n, 1) == 3
        assert [r.to_affine_bytes() for r in mg.finish_batch(1, 3)] == [want[0], want[1], want[0]]
        assert mg.msm(_host(sets[1])).to_affine_bytes() == want[1]  # the synchronous call = one vector through slot 0
    finally:
        mg.close()


def test_config4_2p24_single_gpu(ctx):
    n = 1 << 24
    pts, sc = ctx.sample_points(n, 0xC4_0001), ctx.sample_scalars(n, 0xC4_0002)
    ctx.set_bases(pts)
    whole = ctx.msm(sc)
    # the 8 window ranges an 8-GPU run would take, gathered and combined
    parts = [ctx.msm_windows(sc, *window_range(r, 8)) for r in range(8)]
    assert m.MsmContext.combine_windows(torch.cat(parts, dim=0)) == whole
    # split consistency: MSM(P, s) == MSM(P[:h], s[:h]) + MSM(P[h:], s[h:])
    h = n // 2 + 54321
    first = ctx.msm(sc[:h].contiguous()).to_affine()
    ctx.set_bases(pts[h:].contiguous())
    second = ctx.msm(sc[h:].contiguous()).to_affine()
    assert ref.add(first, second) == whole.to_affine()
    # oracle on a 2^16 slice from the middle of the same inputs
    k, off = 1 << 16, 5 << 20
    ctx.set_bases(pts[off:off + k].contiguous())
    got = ctx.msm(sc[off:off + k].contiguous())
    assert got.to_affine_bytes() == cpu.to_affine64(cpu.cpu_msm(_host(pts[off:off + k]), _host(sc[off:off + k]), THREADS))


def test_config5_batch_64_x_2p18_shared_base(ctx):
    n, batch = 1 << 18, 64
    pts = ctx.sample_points(n, 0xC5_0001)
    sc = ctx.sample_scalars(n * batch, 0xC5_0002)  # 64 independent scalar vectors, contiguous
    ctx.set_bases(pts)
    got = ctx.msm_batch(sc, n)
    assert len(got) == batch
    pb = _host(pts)
    for k in (0, 31, 63):
        assert got[k].to_affine_bytes() == cpu.to_affine64(cpu.cpu_msm(pb, _host(sc[k * n:(k + 1) * n]), THREADS)), k
    assert len({g.to_affine_bytes() for g in got}) == batch  # independent vectors: all results differ
    # the 8-rank partition of the batch (whole MSMs per rank, no exchange on the data path): rank shares in order == whole
    shares = []
    for rank in range(8):
        b, e = batch_range(rank, 8, batch)
        assert e - b == 8
        shares += ctx.msm_batch(sc[b * n:e * n], n)
    assert [g.to_affine_bytes() for g in shares] == [g.to_affine_bytes() for g in got]


def test_configs_2_4_5_with_endomorphism_bases(ctx, inputs_2p20):
    # the same real sizes with MSM_HIP_BASES_ENDOMORPHISM (what bench.py's single-GPU line runs): C2 against the oracle,
    # C5's batch against the plain mode's results and the oracle, C4 against the plain mode (split-consistent above)
    n, pts, sets, want = inputs_2p20
    ctx.set_bases(pts, endomorphism=True)
    for s, w in zip(sets, want):
        assert ctx.msm(s).to_affine_bytes() == w
    ctx.launch_host(_host(sets[0]), 0)
    ctx.launch(sets[1], 1)
    assert ctx.finish(0).to_affine_bytes() == want[0] and ctx.finish(1).to_affine_bytes() == want[1]

    n5, batch = 1 << 18, 64
    pts5 = ctx.sample_points(n5, 0xC5_0001)
    sc5 = ctx.sample_scalars(n5 * batch, 0xC5_0002)
    ctx.set_bases(pts5, endomorphism=True)
    got = ctx.msm_batch(sc5, n5)
    assert got[63].to_affine_bytes() == cpu.to_affine64(cpu.cpu_msm(_host(pts5), _host(sc5[63 * n5:]), THREADS))
    ctx.set_bases(pts5)
    assert [g.to_affine_bytes() for g in ctx.msm_batch(sc5, n5)] == [g.to_affine_bytes() for g in got]
    del pts5, sc5

    n4 = 1 << 24
    pts4, sc4 = ctx.sample_points(n4, 0xC4_0001), ctx.sample_scalars(n4, 0xC4_0002)
    ctx.set_bases(pts4, endomorphism=True)
    whole = ctx.msm(sc4)
    ctx.set_bases(pts4)
    assert ctx.msm(sc4) == whole


@pytest.mark.parametrize("endo", [False, True], ids=["plain", "endomorphism"])
def test_config2_size_with_skewed_and_sparse_scalars(ctx, inputs_2p20, endo):
    # 2^20 points with the scalar distributions that reach the rare paths at full scale: every scalar equal (one bucket of 2^20 entries
    # per window: 24 k pieces stitched by 16 workgroups each), a witness-like vector (40 % zeros, 30 % ones: the device-side chunk length),
    # three distinct values (far-apart heavy buckets: the SMVP's jump across empty slots)
    n, pts, sets, _ = inputs_2p20
    pb = _host(pts)
    s = 0x123456789ABCDEF013579BDF2468ACE0FEDCBA9876543210
    eq = torch.tensor(list(s.to_bytes(32, "little")), dtype=torch.uint8, device=pts.device).repeat(n, 1).contiguous()
    gen = torch.Generator(device=pts.device)
    gen.manual_seed(7)
    sel = torch.rand(n, device=pts.device, generator=gen)
    wit = sets[0].clone()
    wit[sel < 0.7] = 0
    wit[(sel >= 0.4) & (sel < 0.7), 0] = 1
    three = sets[1][:3].repeat((n + 2) // 3, 1)[:n].contiguous()
    ctx.set_bases(pts, endomorphism=endo)
    try:
        for name, sc in (("all equal", eq), ("witness-like", wit), ("three values", three)):
            assert ctx.msm(sc).to_affine_bytes() == cpu.to_affine64(cpu.cpu_msm(pb, _host(sc), THREADS)), name
    finally:
        ctx.set_bases(pts)
