"""Fixed-base precomputation (the reference lists it as future work, README.md): MSM_HIP_BASES_PRECOMPUTE stores
2^(16 w) P_i for every window, whole MSMs then use ONE bucket set.  Same group element as the plain engine and the oracle."""
import pytest
import torch

import msm_webgpu_amd as m
from oracle import bn254_ref as ref
from oracle import cpu
from tests.util import R, case_inputs, golden_cases

pytestmark = pytest.mark.gpu


def test_golden_vectors_with_tables(ctx):
    for case in golden_cases():
        points, scalars = case_inputs(case)
        ctx.set_bases(points, check_on_curve=True, precompute=True)
        assert ctx.msm(scalars).to_affine_bytes().hex() == case["expected_affine"], case["name"]


@pytest.mark.parametrize("n", [1, 257, 5000, 1 << 16, (1 << 18) + 3])
def test_tables_match_oracle_and_plain_engine(ctx, n):
    pts, sc = ctx.sample_points(n, 400 + n), ctx.sample_scalars(n, 401 + n)
    pb, sb = pts.cpu().numpy().tobytes(), sc.cpu().numpy().tobytes()
    want = cpu.to_affine64(cpu.cpu_msm(pb, sb, 8))
    ctx.set_bases(pts, precompute=True)
    assert ctx.msm(sc).to_affine_bytes() == want          # device scalars
    assert ctx.msm(sb).to_affine_bytes() == want          # host scalars
    k = max(1, n // 3)                                    # a prefix of the bases: table stride stays n
    assert ctx.msm(sc[:k].contiguous()).to_affine_bytes() == cpu.to_affine64(cpu.cpu_msm(pb[: 64 * k], sb[: 32 * k], 8))
    # the window-sharding entry points ignore the tables (table 0 is the plain base set)
    parts = [ctx.msm_windows(sc, 0, 7), ctx.msm_windows(sc, 7, 16)]
    assert m.MsmContext.combine_windows(torch.cat(parts, dim=0)).to_affine_bytes() == want
    ctx.set_bases(pts)                                     # back to plain: the flag does not stick
    assert ctx.msm(sc).to_affine_bytes() == want


def test_tables_extreme_and_skewed_scalars(ctx):
    n = 20000
    points = cpu.sample_points(410, n)
    s = 0x0FED_CBA9_8765_4321_0F1E_2D3C_4B5A_6978_8796_A5B4_C3D2_E1F0 % R
    vals = [s] * (n - 8) + [R - 1, 0, 1, 0x8000, (1 << 253) - 1, int("8000" * 15, 16), 2, R - 2]  # one giant bucket + edge digits
    sb = ref.scalars_to_bytes(vals)
    ctx.set_bases(points, precompute=True)
    assert ctx.msm(sb).to_affine_bytes() == cpu.to_affine64(cpu.cpu_msm(points, sb, 8))
    with pytest.raises(m.MsmHipError) as e:
        ctx.msm(b"\xff" * 32)
    assert e.value.code == -4


def test_tables_batches(ctx):
    n, batch = 3000, 11
    pts = ctx.sample_points(n, 420)
    sc = ctx.sample_scalars(n * batch, 421)
    ctx.set_bases(pts, precompute=True)
    got = ctx.msm_batch(sc, n)
    pb = pts.cpu().numpy().tobytes()
    for k in (0, 5, 10):
        assert got[k].to_affine_bytes() == cpu.to_affine64(cpu.cpu_msm(pb, sc[k * n:(k + 1) * n].cpu().numpy().tobytes())), k
    assert [g.to_affine_bytes() for g in ctx.msm_batch(sc.cpu().numpy().tobytes(), n)] == [g.to_affine_bytes() for g in got]
    # one bucket set per MSM: many more whole MSMs fit one launch than with 16 windows each
    assert ctx.batch_group_size(n) == 64
    assert ctx.launch_batch(sc[: 9 * n].contiguous(), n, 2) == 9
    assert [x.to_affine_bytes() for x in ctx.finish_batch(2, 9)] == [x.to_affine_bytes() for x in got[:9]]
    ctx.set_bases(pts)
    assert ctx.batch_group_size(n) <= 4
