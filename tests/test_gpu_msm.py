"""End-to-end MSM parity through the C ABI (≙ the reference's browser tests tests/test_webgpu_msm_cuzk_*.rs and
src/lib.rs:152-167: GPU MSM == cpu_msm), seeded, bit-exact on the canonical 64-byte affine encoding."""
import pytest
import torch

import msm_webgpu_amd as m
from oracle import bn254_ref as ref
from oracle import cpu
from tests.util import R, case_inputs, golden_cases

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", golden_cases(), ids=lambda c: c["name"])
def test_golden_vectors(ctx, case):
    points, scalars = case_inputs(case)
    ctx.set_bases(points, check_on_curve=True)
    got = ctx.msm(scalars)
    assert got.to_affine_bytes().hex() == case["expected_affine"]


@pytest.mark.parametrize("n", [1, 2, 3, 63, 64, 65, 1000, 4096, 65536, 65540])
def test_msm_matches_cpu_msm(ctx, n):
    points, scalars = cpu.sample_points(40 + n, n), cpu.sample_scalars(41 + n, n)
    got = m.run_webgpu_msm(points, scalars)
    assert got.to_affine_bytes() == cpu.to_affine64(cpu.cpu_msm(points, scalars))


def test_empty_input_is_identity(ctx):
    ctx.set_bases(b"")
    assert ctx.msm(b"").is_identity()


def test_prefix_of_bases(ctx):
    # n scalars against the first n of the resident bases
    points, scalars = cpu.sample_points(50, 3000), cpu.sample_scalars(51, 1000)
    ctx.set_bases(points)
    assert ctx.msm(scalars).to_affine_bytes() == cpu.to_affine64(cpu.cpu_msm(points[: 64 * 1000], scalars))


def test_device_resident_inputs_and_samplers(ctx):
    # the HIP samplers emit exactly the oracle's deterministic inputs; device-pointer entry points give the same result
    n = 5000
    pts = ctx.sample_points(n, 77)
    sc = ctx.sample_scalars(n, 78)
    pb, sb = bytes(pts.cpu().numpy().tobytes()), bytes(sc.cpu().numpy().tobytes())
    assert pb == cpu.sample_points(77, n)
    assert sb == cpu.sample_scalars(78, n)
    ctx.set_bases(pts, check_on_curve=True)
    assert ctx.msm(sc).to_affine_bytes() == cpu.to_affine64(cpu.cpu_msm(pb, sb))


def test_async_slots_pipeline(ctx):
    n = 4096
    pts = ctx.sample_points(n, 79)
    s0, s1 = ctx.sample_scalars(n, 80), ctx.sample_scalars(n, 81)
    ctx.set_bases(pts)
    ctx.launch(s0, 0)
    r0 = ctx.finish(0)
    ctx.launch(s1, 1)
    r1 = ctx.finish(1)
    assert r0 == ctx.msm(s0) and r1 == ctx.msm(s1) and r0 != r1
    # a slot must be collected before it is reused
    ctx.launch(s0, 2)
    with pytest.raises(m.MsmHipError) as e:
        ctx.launch(s1, 2)
    assert e.value.code == -8
    assert ctx.finish(2) == r0


def test_window_sharded_equals_whole(ctx):
    # window ranges as 1/2/4/8 GPUs would take them, gathered and combined on the host
    from msm_webgpu_amd.sharding import window_range

    n = 3000
    pts, sc = ctx.sample_points(n, 82), ctx.sample_scalars(n, 83)
    ctx.set_bases(pts)
    whole = ctx.msm(sc)
    for world in (2, 4, 8, 3):
        parts = []
        for rank in range(world):
            b, e = window_range(rank, world)
            parts.append(ctx.msm_windows(sc, b, e))
        assert m.MsmContext.combine_windows(torch.cat(parts, dim=0)) == whole, world


def test_errors_are_codes_not_aborts(ctx):
    with pytest.raises(m.MsmHipError) as e:
        ctx.set_bases((ref.P).to_bytes(32, "little") + (2).to_bytes(32, "little"))  # x = p: non-canonical (≙ utils.rs:20)
    assert e.value.code == -4
    with pytest.raises(m.MsmHipError) as e:
        ctx.set_bases((1).to_bytes(32, "little") + (3).to_bytes(32, "little"), check_on_curve=True)
    assert e.value.code == -5
    ctx.set_bases(ref.points_to_bytes([ref.G]))
    with pytest.raises(m.MsmHipError) as e:
        ctx.msm(b"\xff" * 32)  # scalar 2^256 - 1: recode overflows (test/utils.rs:150-152 panics)
    assert e.value.code == -4
    with pytest.raises(m.MsmHipError) as e:
        ctx.msm(bytes(64))  # more scalars than bases
    assert e.value.code == -2
    with pytest.raises(ValueError):
        m.points_to_bytes([None])  # infinity is not representable (src/lib.rs:58)


def test_linearity_property(ctx):
    # size-independent property: MSM(P, a) + MSM(P, b) == MSM(P, a + b mod r)
    n = 20000
    pts = ctx.sample_points(n, 84)
    a = ref.bytes_to_scalars(cpu.sample_scalars(85, n))
    b = ref.bytes_to_scalars(cpu.sample_scalars(86, n))
    ctx.set_bases(pts)
    ra = ctx.msm(ref.scalars_to_bytes(a)).to_affine()
    rb = ctx.msm(ref.scalars_to_bytes(b)).to_affine()
    rab = ctx.msm(ref.scalars_to_bytes([(x + y) % R for x, y in zip(a, b)])).to_affine()
    assert ref.add(ra, rb) == rab


def test_sharded_pipeline_single_rank(ctx):
    # the asynchronous multi-GPU pipeline degenerates to one rank: same results, two MSMs in flight
    from msm_webgpu_amd.sharding import ShardedMsmPipeline

    n = 6000
    pts = ctx.sample_points(n, 90)
    sets = [ctx.sample_scalars(n, 91 + i) for i in range(3)]
    ctx.set_bases(pts)
    want = [ctx.msm(s) for s in sets]
    pipe = ShardedMsmPipeline(ctx, 0, 1)
    pipe.issue(sets[0])
    pipe.issue(sets[1])
    pipe.issue(sets[2])
    got = [pipe.complete()]
    pipe.issue(sets[0])
    got += [pipe.complete(), pipe.complete(), pipe.complete()]
    assert got == want + [want[0]]


def test_back_to_back_slots_overlap_is_safe(ctx):
    # bucket reduce of MSM i (reduce stream) overlaps sort + SMVP of MSM i+1 (main stream): results must not interfere
    n = 50000
    pts = ctx.sample_points(n, 95)
    sets = [ctx.sample_scalars(n, 96 + i) for i in range(2)]
    ctx.set_bases(pts)
    want = [ctx.msm(s) for s in sets]
    ctx.launch(sets[0], 0)
    for i in range(1, 8):
        ctx.launch(sets[i & 1], i & 1)
        assert ctx.finish((i - 1) & 1) == want[(i - 1) & 1]
    assert ctx.finish(7 & 1) == want[7 & 1]


def test_large_split_consistency(ctx):
    # BASELINE sizes beyond what the oracle finishes in seconds: MSM(P, s) == MSM(P[:h], s[:h]) + MSM(P[h:], s[h:])
    n = 1 << 22
    h = n // 2 + 12345
    pts, sc = ctx.sample_points(n, 120), ctx.sample_scalars(n, 121)
    ctx.set_bases(pts)
    whole = ctx.msm(sc).to_affine()
    first = ctx.msm(sc[:h].contiguous()).to_affine()
    ctx.set_bases(pts[h:].contiguous())
    second = ctx.msm(sc[h:].contiguous()).to_affine()
    assert ref.add(first, second) == whole
    # and against the oracle on a 2^15 slice of the same inputs
    k = 1 << 15
    ctx.set_bases(pts[:k].contiguous())
    pb, sb = pts[:k].cpu().numpy().tobytes(), sc[:k].cpu().numpy().tobytes()
    assert ctx.msm(sc[:k].contiguous()).to_affine_bytes() == cpu.to_affine64(cpu.cpu_msm(pb, sb))


def test_skewed_scalars_many_equal(ctx):
    # every scalar equal: one bucket per window holds all n entries (the stitch walks thousands of chunks)
    n = 1 << 17
    pts = ctx.sample_points(n, 130)
    s = 0x1234_5678_9ABC_DEF0_1357_9BDF_2468_ACE0_FEDC_BA98_7654_3210 % R
    sb = s.to_bytes(32, "little") * n
    ctx.set_bases(pts)
    got = ctx.msm(sb).to_affine()
    ones = (1).to_bytes(32, "little") * n
    total = ctx.msm(ones).to_affine()          # sum of all points (also a single-bucket case)
    assert got == ref.mul(s, total)
    # huge coarse bins get their sub-range histograms from k_fine_hist (above); the fallback -- every sharer of a bin
    # histograms it itself -- must agree
    half = bytearray(sb)
    half[: 32 * (n // 2)] = ctx.sample_scalars(n // 2, 131).cpu().numpy().tobytes()  # half uniform, half equal
    mixed = ctx.msm(bytes(half))
    ctx.set_fine_hist_min_n(1 << 40)
    try:
        assert ctx.msm(sb).to_affine() == got
        assert ctx.msm(bytes(half)) == mixed
    finally:
        ctx.set_fine_hist_min_n(32769)
    pb = pts[:4096].cpu().numpy().tobytes()
    ctx.set_bases(pb)
    assert ctx.msm(ones[: 32 * 4096]).to_affine_bytes() == cpu.to_affine64(cpu.cpu_msm(pb, ones[: 32 * 4096]))


@pytest.mark.parametrize("kind", ["two_bit", "half_equal", "few_values"])
def test_skewed_distributions_match_oracle(ctx, kind):
    # realistic skew (boolean / tiny witness values, repeated values): a few buckets hold most entries, so the
    # big-bucket stitch and the aggregated LDS ranking are exercised; result must stay bit-exact
    n = 20000
    rnd = __import__("random").Random(7)
    base = ref.bytes_to_scalars(cpu.sample_scalars(140, n))
    if kind == "two_bit":
        sc = [rnd.randrange(4) for _ in range(n)]
    elif kind == "half_equal":
        sc = [base[0] if i % 2 else base[i] for i in range(n)]
    else:
        vals = [base[i] for i in range(5)] + [0, 1, R - 1]
        sc = [vals[rnd.randrange(len(vals))] for _ in range(n)]
    points = cpu.sample_points(141, n)
    sb = ref.scalars_to_bytes(sc)
    ctx.set_bases(points)
    assert ctx.msm(sb).to_affine_bytes() == cpu.to_affine64(cpu.cpu_msm(points, sb))


@pytest.mark.parametrize("endo", [False, True], ids=["plain", "endomorphism"])
@pytest.mark.parametrize("kind", ["3_values", "10_values", "20_values", "witness_like", "sparse_5_percent"])
def test_heavy_buckets_and_sparse_vectors_at_2p17(ctx, kind, endo):
    # the paths that only large skewed / sparse inputs reach ("Skewed and sparse scalars"):
    #   buckets of >= 1024 pieces shared by several workgroups (3 values: 16 workgroups per bucket and the last-arrival hand-off;
    #   10 values: more than 128 such buckets, one workgroup each; 20 values: a queue of > 256 items, no sharing),
    #   the SMVP's jump across long gaps of empty slots, and the chunk length settled on the device from the real entry count
    #   (witness-like: 40 % zeros, 30 % ones; 5 % non-zero)
    n = 1 << 17
    rnd = __import__("random").Random(11)
    pts = ctx.sample_points(n, 150)
    base = ref.bytes_to_scalars(ctx.sample_scalars(n, 151).cpu().numpy().tobytes())
    if kind.endswith("_values"):
        vals = base[: int(kind.split("_")[0])]
        sc = [vals[rnd.randrange(len(vals))] for _ in range(n)]
    elif kind == "witness_like":
        sc = [0 if (u := rnd.random()) < 0.4 else 1 if u < 0.7 else base[i] for i in range(n)]
    else:
        sc = [base[i] if rnd.random() < 0.05 else 0 for i in range(n)]
    pb, sb = pts.cpu().numpy().tobytes(), ref.scalars_to_bytes(sc)
    ctx.set_bases(pts, endomorphism=endo)
    try:
        got = ctx.msm(sb)
        assert got.to_affine_bytes() == cpu.to_affine64(cpu.cpu_msm(pb, sb, 8))
        assert ctx.msm(sb) == got  # the arrival counters of the shared buckets are back at zero for the slot's next launch
    finally:
        ctx.set_bases(pts)


def test_batch_over_shared_base(ctx):
    # BASELINE config 5 in miniature: many scalar vectors over one resident base, pipelined inside the library
    n, batch = 3000, 7
    pts = ctx.sample_points(n, 150)
    sc = ctx.sample_scalars(n * batch, 151)
    ctx.set_bases(pts)
    got = ctx.msm_batch(sc, n)
    assert len(got) == batch
    pb = pts.cpu().numpy().tobytes()
    for k in range(batch):
        sb = sc[k * n:(k + 1) * n].cpu().numpy().tobytes()
        assert got[k].to_affine_bytes() == cpu.to_affine64(cpu.cpu_msm(pb, sb)), k
    # the same vectors handed over as host bytes (msm_hip_run_batch_bn254: staged per MSM inside the pipeline)
    host = ctx.msm_batch(sc.cpu().numpy().tobytes(), n)
    assert [g.to_affine_bytes() for g in host] == [g.to_affine_bytes() for g in got]
    # whole-MSM sharding (config 5 on several GPUs), degenerate single-rank case: same results in MSM order
    from msm_webgpu_amd.sharding import sharded_batch_msm

    assert [g.to_affine_bytes() for g in sharded_batch_msm(ctx, sc, n, 0, 1)] == [g.to_affine_bytes() for g in got]
    # a non-canonical scalar in the middle of a batch fails the batch and leaves the context usable (all slots collected)
    bad = bytearray(sc.cpu().numpy().tobytes())
    bad[3 * n * 32:3 * n * 32 + 32] = b"\xff" * 32
    with pytest.raises(m.MsmHipError) as e:
        ctx.msm_batch(bytes(bad), n)
    assert e.value.code == -4
    assert ctx.msm_batch(sc, n)[0].to_affine_bytes() == got[0].to_affine_bytes()


def test_several_msms_per_launch(ctx):
    # msm_hip_launch_windows_batch_device_bn254: the window shares of several MSMs through one kernel sequence
    from msm_webgpu_amd.sharding import ShardedMsmPipeline, msms_per_launch

    assert [msms_per_launch(w) for w in (1, 2, 3, 4, 8, 16)] == [1, 2, 2, 4, 8, 16]
    n = 5000
    pts = ctx.sample_points(n, 170)
    ctx.set_bases(pts)
    vecs = [ctx.sample_scalars(n, 171 + k) for k in range(8)]
    vecs[3] = torch.zeros_like(vecs[3])  # an all-zero vector inside the group: identity
    want = [ctx.msm(v) for v in vecs]
    assert want[3].is_identity()
    batch = torch.cat(vecs, dim=0).contiguous()
    # raw entry point: 8 vectors x windows [6, 8) -> 16 local windows
    out = torch.empty((16, 96), dtype=torch.uint8, device=batch.device)
    ctx.launch_windows_batch(batch, n, 6, 8, 0, out)
    ctx.slot_sync(0)
    for k in (0, 3, 7):  # same group elements (the Jacobian representation depends on the chunking, the point does not)
        ref = ctx.msm_windows(vecs[k], 6, 8)
        for j in range(2):
            assert m.MsmContext.combine_windows(out[2 * k + j:2 * k + j + 1]) == m.MsmContext.combine_windows(ref[j:j + 1]), (k, j)
    with pytest.raises(m.MsmHipError):
        ctx.launch_windows_batch(batch, n, 0, 9, 1, out)  # 8 x 9 local windows exceed the 64 a launch may carry
    # pipeline with groups: single rank doing all 16 windows can only take one MSM per launch ...
    pipe = ShardedMsmPipeline(ctx, 0, 1, msms_per_issue=1)
    pipe.issue(vecs[0])
    assert pipe.complete() == want[0]
    # ... the share of an 8-rank run takes 8: results are the partial sums over windows [0, 2) of every vector
    pipe8 = ShardedMsmPipeline(ctx, 0, 1, msms_per_issue=8, emulate_world=8)
    pipe8.issue(batch, n)
    pipe8.issue(batch[: 3 * n], n)  # a short final group
    got, got3 = pipe8.complete(), pipe8.complete()
    assert len(got) == 8 and len(got3) == 3
    for k in range(8):
        part = m.MsmContext.combine_windows(ctx.msm_windows(vecs[k], 0, 2))
        assert got[k] == part and (k >= 3 or got3[k] == part), k


def test_whole_small_msms_share_a_launch(ctx):
    # up to 4 whole MSMs (64 local windows) per launch: what the batch entry points do for small n
    n = 3000
    pts = ctx.sample_points(n, 180)
    ctx.set_bases(pts)
    vecs = [ctx.sample_scalars(n, 181 + k) for k in range(4)]
    want = [ctx.msm(v) for v in vecs]
    batch = torch.cat(vecs, dim=0).contiguous()
    assert ctx.launch_batch(batch, n, 1) == 4
    assert ctx.finish_batch(1, 4) == want
    assert ctx.launch_batch(batch[: 3 * n], n, 2) == 3
    assert ctx.finish_batch(2, 3) == want[:3]
    with pytest.raises(m.MsmHipError):
        ctx.launch_batch(torch.cat(vecs + vecs[:1], dim=0).contiguous(), n, 3)  # 5 x 16 local windows do not fit
    # the library's own batch runner groups them (9 MSMs -> groups of 4, 4, 1) and still returns them in order
    nine = torch.cat([vecs[k % 4] for k in range(9)], dim=0).contiguous()
    assert ctx.msm_batch(nine, n) == [want[k % 4] for k in range(9)]
    assert ctx.msm(vecs[0]) == want[0]  # single launches still fine afterwards


def test_bases_in_r256_montgomery_form(ctx):
    # MSM_HIP_BASES_MONT256: x * 2^256 mod p words (a 4 x 64-bit Montgomery library's in-memory form) give the same MSM
    n = 777
    points, scalars = cpu.sample_points(190, n), cpu.sample_scalars(191, n)
    pts = ref.bytes_to_points(points)
    P = m.api.P
    mont = b"".join(((x << 256) % P).to_bytes(32, "little") + ((y << 256) % P).to_bytes(32, "little") for x, y in pts)
    want = cpu.to_affine64(cpu.cpu_msm(points, scalars))
    ctx.set_bases(mont, check_on_curve=True, mont256=True)
    assert ctx.msm(scalars).to_affine_bytes() == want
    ctx.set_bases(points, check_on_curve=True)
    assert ctx.msm(scalars).to_affine_bytes() == want
    with pytest.raises(m.MsmHipError) as e:  # canonical bytes read as Montgomery words describe points that are not on the curve
        ctx.set_bases(points, check_on_curve=True, mont256=True)
    assert e.value.code == -5


def test_scalars_in_r256_montgomery_form(ctx):
    # MSM_HIP_SCALARS_MONT256: s * 2^256 mod r words are converted on the device; every entry point sees the same scalars
    n = 1500
    points = cpu.sample_points(192, n)
    sc = ref.bytes_to_scalars(cpu.sample_scalars(193, n))
    sc[:6] = [0, 1, R - 1, R - 2, (R - 1) // 2, 1 << 253]
    canon = ref.scalars_to_bytes(sc)
    mont = b"".join(((s << 256) % R).to_bytes(32, "little") for s in sc)
    want = cpu.to_affine64(cpu.cpu_msm(points, canon))
    ctx.set_bases(points)
    ctx.set_scalar_format(True)
    try:
        assert ctx.msm(mont).to_affine_bytes() == want                                   # host bytes
        t = torch.frombuffer(bytearray(mont + mont), dtype=torch.uint8).cuda()
        assert [g.to_affine_bytes() for g in ctx.msm_batch(t, n)] == [want, want]          # grouped device batch
        bad = bytearray(mont)
        bad[32 * 7:32 * 8] = R.to_bytes(32, "little")                                      # not below r
        with pytest.raises(m.MsmHipError) as e:
            ctx.msm(bytes(bad))
        assert e.value.code == -4
    finally:
        ctx.set_scalar_format(False)
    assert ctx.msm(canon).to_affine_bytes() == want


def test_inputs_produced_asynchronously_on_the_torch_stream(ctx):
    # the engine's streams are not ordered with torch's: every launch wrapper makes the engine wait for the producing torch
    # stream (msm_hip_wait_stream), and a slot keeps its temporaries alive until it is collected
    n = 1 << 18
    pts = ctx.sample_points(n, 200)
    base = [ctx.sample_scalars(n, 201 + k) for k in range(3)]
    ctx.set_bases(pts)
    want = [ctx.msm(b) for b in base]
    filler = torch.empty((64 << 20,), dtype=torch.uint8, device=pts.device)
    for rep in range(3):
        for k in range(3):
            filler.fill_(rep)                                   # keeps the torch stream busy ahead of the producer
            t = torch.cat([base[k][: n // 2], base[k][n // 2:]], dim=0)  # fresh temporary, produced asynchronously
            ctx.launch(t, k)
            del t                                                # the caching allocator may hand the block out again
            junk = torch.full((n, 32), 0xFF, dtype=torch.uint8, device=pts.device)  # ... to this (non-canonical scalars)
            del junk
        assert [ctx.finish(k) for k in range(3)] == want


def test_one_shot_keeps_its_context_between_calls(built):
    # msm_hip_msm_bn254_g1 ≙ compute_msm as the reference calls it (src/cuzk/msm.rs:75-94); the library keeps the context it used:
    # different inputs and sizes (growing and shrinking) through the kept context, release, again
    import ctypes as C

    L = m.lib()
    out = C.create_string_buffer(96)
    for round_ in range(2):
        for n, seed in ((300, 1), (5000, 2), (70000, 3), (17, 4), (5000, 2)):
            points, scalars = cpu.sample_points(700 + seed, n), cpu.sample_scalars(800 + seed, n)
            assert L.msm_hip_msm_bn254_g1(points, scalars, n, out) == 0
            assert cpu.to_affine64(out.raw) == cpu.to_affine64(cpu.cpu_msm(points, scalars)), (round_, n)
        # an input error leaves the kept context usable
        assert L.msm_hip_msm_bn254_g1(cpu.sample_points(1, 2), b"\xff" * 64, 2, out) == -4
        assert L.msm_hip_msm_bn254_g1(cpu.sample_points(1, 2), bytes(64), 2, out) == 0 and cpu.to_affine64(out.raw) == bytes(64)
        L.msm_hip_oneshot_release()
    L.msm_hip_oneshot_release()  # nothing kept: a no-op


def test_one_shot_in_parts(built):
    """Round 5: from 2^19 points on the one-shot call (≙ compute_msm, src/cuzk/msm.rs:75-94) runs as sub-MSMs over ranges of the points, a context each,
    their uploads one behind the other, and adds the results on the host.  The test hook lowers the threshold so that the oracle checks every
    part count on small inputs: ragged ranges, fewer points than parts, an input error in a LATER part (the earlier ones are drained, the kept
    contexts stay usable), another curve, and the real threshold at 2^19 + 3 against the persistent context's result."""
    import ctypes as C

    from oracle import cpu_grumpkin as gcpu

    L = m.lib()
    out = C.create_string_buffer(96)
    try:
        for parts in (2, 3, 4):
            assert L.msm_hip_test_oneshot_parts(parts, 1) == 0
            for n, seed in ((1, 1), (3, 2), (5, 3), (301, 4), (5000, 5), (70001, 6), (302, 7)):
                points, scalars = cpu.sample_points(900 + seed, n), cpu.sample_scalars(950 + seed, n)
                assert L.msm_hip_msm_bn254_g1(points, scalars, n, out) == 0
                assert cpu.to_affine64(out.raw) == cpu.to_affine64(cpu.cpu_msm(points, scalars)), (parts, n)
            # a non-canonical scalar in the last part, a non-canonical coordinate in the first: the call fails as a whole, the next one works
            n = 1000
            points, scalars = cpu.sample_points(31, n), cpu.sample_scalars(32, n)
            bad_sc = scalars[:-32] + b"\xff" * 32
            assert L.msm_hip_msm_bn254_g1(points, bad_sc, n, out) == -4
            bad_pt = b"\xff" * 32 + points[32:]
            assert L.msm_hip_msm_bn254_g1(bad_pt, scalars, n, out) == -4
            assert L.msm_hip_msm_bn254_g1(points, scalars, n, out) == 0
            assert cpu.to_affine64(out.raw) == cpu.to_affine64(cpu.cpu_msm(points, scalars))
            # opposite halves: part 1 = -(part 0) gives the identity
            half = cpu.sample_points(33, 500)
            neg = b"".join(half[64 * k:64 * k + 32] + ((cpu.constants()["p"] - int.from_bytes(half[64 * k + 32:64 * k + 64], "little")) % cpu.constants()["p"]).to_bytes(32, "little") for k in range(500))
            sc = cpu.sample_scalars(34, 500)
            if parts == 2:
                assert L.msm_hip_msm_bn254_g1(half + neg, sc + sc, 1000, out) == 0 and cpu.to_affine64(out.raw) == bytes(64)
        # another curve through the curve-neutral entry point
        assert L.msm_hip_test_oneshot_parts(3, 1) == 0
        gp, gs = gcpu.sample_points(41, 2001), gcpu.sample_scalars(42, 2001)
        assert L.msm_hip_msm_curve(1, gp, gs, 2001, out) == 0 and gcpu.to_affine64(out.raw) == gcpu.to_affine64(gcpu.cpu_msm(gp, gs))
    finally:
        assert L.msm_hip_test_oneshot_parts(0, 0) == 0
        L.msm_hip_oneshot_release()
    # the default policy: two parts at 2^19 + 3 points; compared with the same MSM on a persistent context (one part by construction)
    n = (1 << 19) + 3
    c = m.MsmContext(0)
    try:
        pts, sc = c.sample_points(n, 77), c.sample_scalars(n, 78)
        c.set_bases(pts, endomorphism=None)
        want = c.msm(sc).to_affine()
        pb, sb = bytes(pts.cpu().numpy().tobytes()), bytes(sc.cpu().numpy().tobytes())
    finally:
        c.close()
    assert L.msm_hip_msm_bn254_g1(pb, sb, n, out) == 0
    assert m.G1(out.raw).to_affine() == want
    L.msm_hip_oneshot_release()


def test_host_scalar_run_in_parts(built):
    """msm_hip_run with host scalars (scope B: resident bases, 32 n bytes over the host link per call) runs from 2^19 points on as sub-MSMs over ranges
    of the points, one result slot each, so that a part's scalars arrive while the previous part is accumulated.  With the threshold lowered by the
    test hook: every part count, ragged ranges, fewer scalars than bases, both base modes, a bad scalar in a later part, a busy slot (falls back to
    the one-part path), and the default policy at 2^20 + 5 against the device-scalar path (one part by construction)."""
    L = m.lib()
    n = 6001
    points, scalars = cpu.sample_points(61, n), cpu.sample_scalars(62, n)
    c = m.MsmContext(0)
    try:
        for endo in (None, False):
            c.set_bases(points, endomorphism=endo)
            for parts in (2, 3, 4):
                assert L.msm_hip_test_oneshot_parts(parts, 1) == 0
                for k in (n, 4097, 7, 2):
                    assert c.msm(scalars[:32 * k]).to_affine_bytes() == cpu.to_affine64(cpu.cpu_msm(points[:64 * k], scalars[:32 * k])), (endo, parts, k)
                with pytest.raises(m.MsmHipError):
                    c.msm(scalars[:-32] + b"\xff" * 32)
                assert c.msm(scalars).to_affine_bytes() == cpu.to_affine64(cpu.cpu_msm(points, scalars))
            # a launch of the caller's own in slot 1: the call takes the one-part path and leaves that launch alone
            assert L.msm_hip_test_oneshot_parts(2, 1) == 0
            c.launch_host(scalars[:32 * 100], 1)
            assert c.msm(scalars).to_affine_bytes() == cpu.to_affine64(cpu.cpu_msm(points, scalars))
            assert c.finish(1).to_affine_bytes() == cpu.to_affine64(cpu.cpu_msm(points[:64 * 100], scalars[:32 * 100]))
    finally:
        assert L.msm_hip_test_oneshot_parts(0, 0) == 0
        c.close()
    n = (1 << 20) + 5
    c = m.MsmContext(0)
    try:
        pts, sc = c.sample_points(n, 81), c.sample_scalars(n, 82)
        c.set_bases(pts, endomorphism=None)
        want = c.msm(sc)                                   # device scalars: one launch
        assert c.msm(bytes(sc.cpu().numpy().tobytes())) == want   # host scalars: two parts by the default policy
    finally:
        c.close()
