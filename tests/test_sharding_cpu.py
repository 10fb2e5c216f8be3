"""The multi-GPU path on CPU: window partition + one all-gather (gloo, world_size 2) + host combine.
Per-rank window sums come from the oracle's stage models here (no GPU); the collective and the combine are the
product's (msm-webgpu_amd/sharding.py, msm_hip_combine_windows_bn254)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import cpu


def test_window_range_partitions():
    from msm_webgpu_amd.sharding import max_windows_per_rank, window_range

    for world in (1, 2, 3, 4, 5, 8, 16):
        ranges = [window_range(r, world) for r in range(world)]
        assert ranges[0][0] == 0 and ranges[-1][1] == 16
        assert all(ranges[i][1] == ranges[i + 1][0] for i in range(world - 1))
        assert max(e - b for b, e in ranges) == max_windows_per_rank(world)
        assert max(e - b for b, e in ranges) - min(e - b for b, e in ranges) <= 1
    with pytest.raises(ValueError):
        window_range(2, 2)


def _oracle(curve):
    import importlib

    return cpu if curve == "bn254" else importlib.import_module("oracle.cpu_" + curve)


def _window_sums(c, points, scalars, b, e):
    digits = c.decompose_scalars_signed(scalars)
    out = []
    for w in range(b, e):
        cp, vi = c.transpose(digits[w], 1 << 16)
        out.append(c.bucket_reduction("running_sum", c.smvp_signed(cp, vi, points, 1 << 16)))
    return b"".join(out)


def _worker(rank, world, port, n, q, curve="bn254"):
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import msm_webgpu_amd as m
    from msm_webgpu_amd.sharding import gather_window_sums, window_range

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    c = _oracle(curve)
    jb = 3 * c.coord_bytes()  # bytes of a Jacobian record on this curve
    points, scalars = c.sample_points(91, n), c.sample_scalars(92, n)
    b, e = window_range(rank, world)
    local = torch.from_numpy(np.frombuffer(_window_sums(c, points, scalars, b, e), dtype=np.uint8).copy()).view(e - b, jb)
    all_sums = gather_window_sums(local, rank, world, jb=jb)
    result = m.MsmContext.combine_windows(all_sums, curve=curve)
    q.put((rank, result.to_affine_bytes()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,curve", [(2, "bn254"), (3, "bn254"), (2, "bls12_381"), (2, "bn254_g2")])
def test_sharded_gather_and_combine_gloo(built, world, curve):
    # (the other curves: 144-byte records on BLS12-381, 192-byte ones on G2 -- coordinates in Fq2 -- through the same gather and combine)
    n = 600 if curve == "bn254" else 150
    cpu = _oracle(curve)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q, curve)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = cpu.to_affine64(cpu.cpu_msm(cpu.sample_points(91, n), cpu.sample_scalars(92, n)))
    assert all(g[1] == want for g in got)


def _batch_worker(rank, world, port, n, batch, q):
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from msm_webgpu_amd.sharding import batch_range, gather_batch_results

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    points = cpu.sample_points(71, n)
    b, e = batch_range(rank, world, batch)
    mine = b"".join(cpu.cpu_msm(points, cpu.sample_scalars(500 + k, n)) for k in range(b, e))  # this rank's whole MSMs (oracle)
    local = torch.from_numpy(np.frombuffer(mine, dtype=np.uint8).copy()).view(e - b, 96)
    allr = gather_batch_results(local, rank, world, batch)
    q.put((rank, allr.numpy().tobytes()))
    dist.barrier()
    dist.destroy_process_group()


def test_batch_sharding_gloo(built):
    # BASELINE config 5 shape: whole MSMs sharded over ranks, results gathered in MSM order on every rank
    from msm_webgpu_amd.sharding import batch_range

    assert [batch_range(r, 8, 64) for r in (0, 7)] == [(0, 8), (56, 64)]
    world, n, batch = 2, 200, 5
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_batch_worker, args=(r, world, port, n, batch, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    points = cpu.sample_points(71, n)
    want = [cpu.to_affine64(cpu.cpu_msm(points, cpu.sample_scalars(500 + k, n))) for k in range(batch)]
    for _, blob in got:
        assert [cpu.to_affine64(blob[96 * k:96 * k + 96]) for k in range(batch)] == want


def _group_worker(rank, world, port, nvec, q):
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from msm_webgpu_amd.sharding import group_window_rows, max_windows_per_rank, window_range

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    per = max_windows_per_rank(world)
    b, e = window_range(rank, world)
    # what a rank's launch writes: [nvec][w_local] records, vector-major; record (v, w) is tagged with bytes v, w
    padded = torch.zeros((nvec * per, 96), dtype=torch.uint8)
    for v in range(nvec):
        for w in range(b, e):
            padded[v * (e - b) + (w - b)] = torch.tensor([v, w] + [0] * 94, dtype=torch.uint8)
    gathered = torch.empty((world, nvec * per, 96), dtype=torch.uint8)
    dist.all_gather_into_tensor(gathered.view(-1), padded.view(-1))
    ok = all(group_window_rows(gathered, v, world)[:, :2].tolist() == [[v, w] for w in range(16)] for v in range(nvec))
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


def test_grouped_launch_layout_gloo():
    # several MSMs per launch: the all-gathered block of every rank is vector-major; uneven window counts (3 ranks: 6, 5, 5)
    from msm_webgpu_amd.sharding import msms_per_launch

    world = 3
    nvec = msms_per_launch(world)
    assert nvec == 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_group_worker, args=(r, world, port, nvec, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in got)


@pytest.mark.parametrize("world,num_windows", [(1, 16), (2, 16), (3, 16), (8, 16), (16, 16), (2, 8), (3, 8), (8, 8)])
def test_gathered_blocks_to_per_msm_window_sums(built, world, num_windows):
    """The layout step of ShardedMsmPipeline.complete (16 full-length windows, or the 8 half-length ones of endomorphism bases) and the
    one-call batch combine behind it (msm_hip_combine_windows_batch_curve, host only)."""
    import msm_webgpu_amd as m
    from msm_webgpu_amd.sharding import gathered_window_sums, max_windows_per_rank, window_range

    nvec = 3
    per = max_windows_per_rank(world, num_windows)
    recs = cpu.g1_scalar_mul(cpu.sample_points(5, nvec * num_windows), cpu.sample_scalars(6, nvec * num_windows))  # arbitrary group elements
    rec = lambda v, w: recs[96 * (v * num_windows + w):96 * (v * num_windows + w) + 96]
    host = np.zeros((world, nvec * per + 2, 96), dtype=np.uint8)  # (+2: a block may be longer than what the launch filled)
    for r in range(world):
        b, e = window_range(r, world, num_windows)
        for v in range(nvec):
            for w in range(b, e):
                host[r, v * (e - b) + (w - b)] = np.frombuffer(rec(v, w), dtype=np.uint8)
    sums = gathered_window_sums(host, nvec, world, num_windows)
    assert sums.shape == (nvec, num_windows, 96) and sums.tobytes() == recs
    got = m.MsmContext.combine_windows_batch(sums, num_windows)
    for v in range(nvec):
        want = cpu.horner(recs[96 * num_windows * v:96 * num_windows * (v + 1)])
        assert got[v].to_affine_bytes() == cpu.to_affine64(want)
        assert m.MsmContext.combine_windows(recs[96 * num_windows * v:96 * num_windows * (v + 1)]).to_affine_bytes() == cpu.to_affine64(want)
    assert m.MsmContext.combine_windows_batch(b"", num_windows) == []


@pytest.mark.parametrize("mode", ["all", "spread", "rank0"])
def test_pipeline_combines_every_msm_once_across_the_ranks(built, mode):
    """ShardedMsmPipeline's host combine (after the all-gather every rank holds all window sums): "spread" -- vector v by rank v % world --
    and "rank0" run each Horner chain ONCE across the ranks instead of once per rank; the owners' results are the oracle's.  Host only:
    the object is filled in by hand (its constructor allocates device buffers)."""
    from msm_webgpu_amd.sharding import ShardedMsmPipeline, max_windows_per_rank, window_range

    world, nvec, nw = 3, 4, 16
    per = max_windows_per_rank(world, nw)
    recs = cpu.g1_scalar_mul(cpu.sample_points(15, nvec * nw), cpu.sample_scalars(16, nvec * nw))
    host = np.zeros((world, nvec * per, 96), dtype=np.uint8)
    for r in range(world):
        b, e = window_range(r, world, nw)
        for v in range(nvec):
            for w in range(b, e):
                host[r, v * (e - b) + (w - b)] = np.frombuffer(recs[96 * (v * nw + w):96 * (v * nw + w) + 96], dtype=np.uint8)
    want = [cpu.to_affine64(cpu.horner(recs[96 * nw * v:96 * nw * (v + 1)])) for v in range(nvec)]

    class Ctx:
        curve = "bn254"

    owners = []
    for rank in range(world):
        p = ShardedMsmPipeline.__new__(ShardedMsmPipeline)
        p.combine, p.emulate, p.world, p.rank, p.num_windows, p.ctx, p.wide = mode, 0, world, rank, nw, Ctx(), False
        p.w_begin, p.w_end = window_range(rank, world, nw)
        p.host_np = [host]
        out = p._combine(0, nvec)
        assert len(out) == nvec
        for v, g in enumerate(out):
            if g is not None:
                owners.append(v)
                assert g.to_affine_bytes() == want[v]
            else:
                assert p.owner(v) != rank
    assert sorted(owners) == (sorted(list(range(nvec)) * world) if mode == "all" else list(range(nvec)))


def _pairs_worker(rank, world, port, nvec, nwin, q):
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from msm_webgpu_amd.sharding import gathered_window_sums, max_windows_per_rank, window_range

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    b, e = window_range(rank, world, nwin)
    per = max_windows_per_rank(world, nwin)
    # this rank's block as msm_hip_launch_vwindows_batch_device writes it: [nvec][its virtual windows][2] records, then padding
    block = torch.zeros((nvec * per * 2, 96), dtype=torch.uint8)
    block[: nvec * (e - b) * 2] = torch.from_numpy(_PAIRS(nvec, nwin)[:, b:e].reshape(-1, 96).copy())
    gathered = torch.empty((world, nvec * per * 2, 96), dtype=torch.uint8)
    dist.all_gather_into_tensor(gathered.view(-1), block.view(-1))
    pairs = gathered_window_sums(gathered.numpy().reshape(world, nvec * per, 192), nvec, world, nwin)
    q.put((rank, pairs.tobytes()))
    dist.barrier()
    dist.destroy_process_group()


def _PAIRS(nvec, nwin):
    """(W_hi, TC_hi) records of `nvec` synthetic MSMs: multiples of the generator, so that the finish has a closed form"""
    from oracle import bn254_ref as ref

    out = np.zeros((nvec, nwin, 2, 96), dtype=np.uint8)
    for v in range(nvec):
        for hi in range(nwin):
            for k, mult in enumerate((1000 * v + 7 * hi + 1, 13 * v + hi + 2)):
                x, y = ref.mul(mult, ref.G)
                out[v, hi, k] = np.frombuffer(x.to_bytes(32, "little") + y.to_bytes(32, "little") + (1).to_bytes(32, "little"), dtype=np.uint8)
    return out


@pytest.mark.parametrize("world,nwin", [(2, 8), (3, 8), (2, 16)])
def test_virtual_window_pairs_gather_and_finish_gloo(built, world, nwin):
    """the record shape of window-sharded launches over wide tables: every rank contributes (weighted sum, plain total) pairs of its virtual
    windows; after ONE all-gather (gloo here) every rank finishes V sum_vw W_vw - sum_vw (V - 1 - vw) TC_vw (msm_hip_combine_vwindows_batch_curve)"""
    import msm_webgpu_amd as m
    from oracle import bn254_ref as ref

    nvec = 3
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_pairs_worker, args=(r, world, port, nvec, nwin, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(g[1] == _PAIRS(nvec, nwin).tobytes() for g in got)  # every rank holds every pair, in virtual-window order
    res = m.MsmContext.combine_vwindows_batch(np.frombuffer(got[0][1], dtype=np.uint8), nwin)
    for v in range(nvec):
        mult = nwin * sum(1000 * v + 7 * hi + 1 for hi in range(nwin)) - sum((nwin - 1 - hi) * (13 * v + hi + 2) for hi in range(nwin))
        x, y = ref.mul(mult % ref.R, ref.G)
        assert res[v].to_affine_bytes() == x.to_bytes(32, "little") + y.to_bytes(32, "little"), v
