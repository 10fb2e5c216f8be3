"""msm_hip_mgpu_*: several engine contexts driven by one process behind the C ABI (windows sharded + gather + ONE host
combine; batches dealt out as whole MSMs).  The GPU box has one GPU, so device ids repeat (pinned-buffer gather) or the list
has one entry (RCCL all-gather with a single rank); the 8-device case is the driver's to run."""
import pytest

import msm_webgpu_amd as m
from oracle import cpu

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def data(ctx):
    n = 50000
    points, scalars = cpu.sample_points(900, n), cpu.sample_scalars(901, n)
    return n, points, scalars, cpu.to_affine64(cpu.cpu_msm(points, scalars, 8))


@pytest.mark.parametrize("ids,gather", [([0], "host"), ([0, 0], "host"), ([0, 0, 0], "host"), ([0] * 8, "host"), ([0], "rccl"), ([0], "auto")])
def test_mgpu_window_sharded_msm(data, ids, gather):
    n, points, scalars, want = data
    mg = m.MultiGpuMsm(ids, gather)
    try:
        assert mg.device_count == len(ids)
        assert mg.uses_rccl == (gather == "rccl")
        assert mg.set_bases(points, check_on_curve=True) == n
        assert mg.msm(scalars).to_affine_bytes() == want
        assert mg.msm(scalars[: 32 * 1000]).to_affine_bytes() == cpu.to_affine64(cpu.cpu_msm(points[: 64 * 1000], scalars[: 32 * 1000]))
        assert mg.msm(b"").is_identity()
        with pytest.raises(m.MsmHipError) as e:  # a non-canonical scalar: every device reports it, the call returns the code
            mg.msm(scalars[:32] + b"\xff" * 32)
        assert e.value.code == -4
        assert mg.msm(scalars).to_affine_bytes() == want  # still usable
    finally:
        mg.close()


def test_mgpu_batch_of_whole_msms(data):
    n, points, scalars, _ = data
    k, batch = 3000, 7
    vecs = cpu.sample_scalars(902, k * batch)
    mg = m.MultiGpuMsm([0, 0, 0], "host")
    try:
        mg.set_bases(points[: 64 * k])
        got = mg.msm_batch(vecs, k)
        assert len(got) == batch
        for j in range(batch):
            assert got[j].to_affine_bytes() == cpu.to_affine64(cpu.cpu_msm(points[: 64 * k], vecs[32 * k * j:32 * k * (j + 1)])), j
        assert mg.msm_batch(b"", k) == []
        # endomorphism bases on every device: the batch path uses them, the window-sharded msm() runs the plain 16 windows
        mg.set_bases(points[: 64 * k], endomorphism=True)
        assert [g.to_affine_bytes() for g in mg.msm_batch(vecs, k)] == [g.to_affine_bytes() for g in got]
        assert mg.msm(vecs[: 32 * k]).to_affine_bytes() == got[0].to_affine_bytes()
    finally:
        mg.close()


@pytest.mark.parametrize("ids,gather,endo", [([0], "rccl", False), ([0], "rccl", True), ([0, 0, 0], "host", False), ([0] * 5, "host", True),
                                             ([0] * 16, "host", True)])
def test_mgpu_grouped_asynchronous_launches(data, ids, gather, endo):
    """msm_hip_mgpu_launch_batch_* / finish_batch: several MSMs' window shares per launch, slots in flight, errors reported by finish
    and every slot left free; uneven shares (3 and 5 devices), more devices than half-length windows (16 > 8: some devices idle)."""
    n, points, scalars, want = data
    other = cpu.sample_scalars(903, n)
    want_other = cpu.to_affine64(cpu.cpu_msm(points, other, 8))
    mg = m.MultiGpuMsm(ids, gather)
    try:
        mg.set_bases(points, endomorphism=endo)
        g = mg.group_size
        assert g == max(1, (8 if endo else 16) // -(-(8 if endo else 16) // len(ids)))
        nv = min(g, 3)
        vecs = (scalars, other, scalars)[:nv]
        wants = (want, want_other, want)[:nv]
        for k in range(4):
            assert mg.launch_batch(b"".join(vecs), n, k) == nv
        with pytest.raises(m.MsmHipError) as e:  # slot still in flight
            mg.launch_batch(scalars, n, 2)
        assert e.value.code == -8
        for k in (2, 0, 3, 1):  # any order
            assert [r.to_affine_bytes() for r in mg.finish_batch(k, nv)] == list(wants)
        with pytest.raises(m.MsmHipError):  # nothing pending
            mg.finish_batch(0, 1)
        # a non-canonical scalar in one vector: finish reports it, the slot and every context stay usable
        mg.launch_batch(scalars[:32] + b"\xff" * 32, 2, 1)
        mg.launch_batch(scalars, n, 2)
        with pytest.raises(m.MsmHipError) as e:
            mg.finish_batch(1, 1)
        assert e.value.code == -4
        assert mg.finish_batch(2, 1)[0].to_affine_bytes() == want
        mg.launch_batch(other, n, 1)
        assert mg.finish_batch(1, 1)[0].to_affine_bytes() == want_other
    finally:
        mg.close()


@pytest.mark.parametrize("wide", [False, True], ids=["windows", "virtual_windows"])
@pytest.mark.parametrize("ids,gather,fault", [([0], "rccl", 0), ([0, 0, 0], "host", 1), ([0] * 8, "host", 7)])
def test_mgpu_a_failing_device_is_reported_within_a_bound_and_the_next_launch_succeeds(data, ids, gather, fault, wide):
    """One device's launch fails (msm_hip_mgpu_inject_fault: before anything is queued there, as a busy slot, an allocation failure or a HIP
    error would).  RCCL gather: that device still issues its call of the launch's all-gather (zeroed block), so the collective completes,
    finish returns the error in bounded time, launches already in flight behind it are unaffected and the next launch succeeds.  (With one
    rank the collective cannot hang -- what this checks is the lock-step bookkeeping; two real GPUs: the test below.)"""
    import time

    n, points, scalars, want = data
    mg = m.MultiGpuMsm(ids, gather)
    try:
        mg.set_bases(points, precompute="wide" if wide else False)   # (wide: the shares are virtual windows, the gathered records come in pairs)
        mg.launch_batch(scalars, n, 0)
        assert mg.finish_batch(0, 1)[0].to_affine_bytes() == want
        mg.inject_fault(fault, 1)
        t0 = time.monotonic()
        mg.launch_batch(scalars, n, 1)   # fails on device `fault`
        mg.launch_batch(scalars, n, 2)   # queued behind it: must be unaffected
        with pytest.raises(m.MsmHipError) as e:
            mg.finish_batch(1, 1)
        assert e.value.code == -7 and time.monotonic() - t0 < 10.0
        assert mg.finish_batch(2, 1)[0].to_affine_bytes() == want
        mg.launch_batch(scalars, n, 1)   # the failed launch's slot is free again
        assert mg.finish_batch(1, 1)[0].to_affine_bytes() == want
        assert mg.msm(scalars).to_affine_bytes() == want
    finally:
        mg.close()


def test_mgpu_two_distinct_devices_over_rccl(data):
    """The in-process RCCL gather on two real GPUs (skipped on a one-GPU box; the driver's multi-GPU node runs it)."""
    import torch

    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    n, points, scalars, want = data
    for endo, wide in ((False, False), (True, False), (False, True)):   # 16 windows, 8 half-length windows, the virtual windows of wide tables (pairs of records)
        mg = m.MultiGpuMsm([0, 1], "rccl")
        try:
            assert mg.uses_rccl
            mg.set_bases(points, endomorphism=endo, precompute="wide" if wide else False)
            assert mg.msm(scalars).to_affine_bytes() == want
            g = mg.group_size
            for k in range(3):
                mg.launch_batch(scalars * g, n, k)
            for k in range(3):
                assert all(r.to_affine_bytes() == want for r in mg.finish_batch(k, g))
            # a failing device must not hang its peer's collective, nor shift the pairing of the launches behind it
            import time

            mg.inject_fault(1, 1)
            t0 = time.monotonic()
            mg.launch_batch(scalars * g, n, 0)
            mg.launch_batch(scalars * g, n, 1)
            with pytest.raises(m.MsmHipError):
                mg.finish_batch(0, g)
            assert time.monotonic() - t0 < 20.0
            assert all(r.to_affine_bytes() == want for r in mg.finish_batch(1, g))
        finally:
            mg.close()


def test_mgpu_bad_arguments():
    with pytest.raises(m.MsmHipError) as e:
        m.MultiGpuMsm([])
    assert e.value.code == -2
    with pytest.raises(m.MsmHipError) as e:
        m.MultiGpuMsm([0] * 17)
    assert e.value.code == -2
    with pytest.raises(m.MsmHipError):
        m.MultiGpuMsm([0, 0], "rccl")  # RCCL cannot put two ranks on one device
