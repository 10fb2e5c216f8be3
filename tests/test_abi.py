"""The C-ABI library builds, loads, and exports every symbol include/msm_hip.h declares; host-only entry points work
without a GPU; device entry points fail loudly (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest
import torch

from oracle import cpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "msm_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(msm_hip_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(built):
    import msm_webgpu_amd as m

    L = m.lib()
    names = _declared()
    assert len(names) >= 25
    for name in names:
        assert hasattr(L, name), name
    assert L.msm_hip_abi_version() == 7
    assert b"ok" == L.msm_hip_strerror(0)


def test_no_gpu_means_loud_failure(built):
    import msm_webgpu_amd as m

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(m.MsmHipError) as e:
        m.MsmContext(0)
    assert e.value.code == -1
    out = C.create_string_buffer(96)
    assert m.lib().msm_hip_msm_bn254_g1(bytes(64), bytes(32), 1, out) == -1
    with pytest.raises(m.MsmHipError):
        m.run_webgpu_msm(bytes(64), bytes(32))


def test_host_window_combine_matches_oracle_horner(built):
    # msm_hip_combine_windows_bn254 is the host finalisation (src/cuzk/msm.rs:411-416); no device involved
    import msm_webgpu_amd as m

    n = 16
    sums = cpu.g1_scalar_mul(cpu.sample_points(3, n), cpu.sample_scalars(4, n))  # 16 arbitrary Jacobian points
    got = m.MsmContext.combine_windows(sums)
    assert got.to_affine_bytes() == cpu.to_affine64(cpu.horner(sums))
    ident = bytes(96 * 16)
    assert m.MsmContext.combine_windows(ident).is_identity()
    out = C.create_string_buffer(96)
    assert m.lib().msm_hip_combine_windows_bn254(b"\xff" * 96, 1, out) == -4  # non-canonical coordinate
    assert m.lib().msm_hip_combine_windows_bn254(ident, 17, out) == -2


def test_wire_format_helpers(built):
    import msm_webgpu_amd as m
    from oracle import bn254_ref as ref

    pts = ref.sample_points(1, 3)
    assert m.points_to_bytes(pts) == ref.points_to_bytes(pts)
    assert m.scalars_to_bytes([1, 2, ref.R - 1]) == ref.scalars_to_bytes([1, 2, ref.R - 1])
    g = m.G1(ref.points_to_bytes([pts[0]]) + (1).to_bytes(32, "little"))
    assert g.to_affine() == pts[0] and not g.is_identity()
    assert m.G1(bytes(96)).to_affine() is None and m.G1(bytes(96)).to_affine_bytes() == bytes(64)


def test_host_to_affine_helper_matches_oracle(built):
    import msm_webgpu_amd as m

    L = m.lib()
    L.msm_hip_g1_to_affine_bn254.argtypes = [C.c_char_p, C.c_char_p]
    pts = cpu.g1_scalar_mul(cpu.sample_points(5, 8), cpu.sample_scalars(6, 8))  # 8 Jacobian points with z != 1
    for k in range(8):
        out = C.create_string_buffer(64)
        assert L.msm_hip_g1_to_affine_bn254(pts[96 * k:96 * k + 96], out) == 0
        assert out.raw == cpu.to_affine64(pts[96 * k:96 * k + 96])
    out = C.create_string_buffer(64)
    assert L.msm_hip_g1_to_affine_bn254(bytes(96), out) == 1 and out.raw == bytes(64)
    assert L.msm_hip_g1_to_affine_bn254(b"\xff" * 96, out) == -4


def test_abi_window_partition_equals_the_python_harness(built):
    # msm_hip_window_range (what msm_hip_mgpu_* shards by) == sharding.window_range (what the torch.distributed harness shards by)
    from msm_webgpu_amd.api import window_range_abi
    from msm_webgpu_amd.sharding import batch_range, window_range

    for world in range(1, 17):
        ranges = [window_range_abi(r, world) for r in range(world)]
        assert ranges == [window_range(r, world) for r in range(world)]
        assert ranges[0][0] == 0 and ranges[-1][1] == 16 and all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
    for batch in (0, 1, 7, 64, 1000):
        assert [window_range_abi(r, 8, batch) for r in range(8)] == [batch_range(r, 8, batch) for r in range(8)]


def test_mgpu_without_gpu_fails_loudly(built):
    import msm_webgpu_amd as m

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(m.MsmHipError) as e:
        m.MultiGpuMsm([0])
    assert e.value.code == -1


def test_host_only_helpers_of_window_sizes_and_curves(built):
    import msm_webgpu_amd as m
    from oracle import cpu_grumpkin as cg

    L = m.lib()
    nw, nb = C.c_int(), C.c_int()
    assert [(L.msm_hip_window_config(b, C.byref(nw), C.byref(nb)), nw.value, nb.value) for b in (12, 14, 16)] == [(0, 22, 2048), (0, 19, 8192), (0, 16, 32768)]
    assert L.msm_hip_window_config(13, C.byref(nw), C.byref(nb)) == -2
    # the host window combine for the second curve (no device involved): Horner over 16 arbitrary Grumpkin points
    sums = cg.g1_scalar_mul(cg.sample_points(3, 16), cg.sample_scalars(4, 16))
    got = m.MsmContext.combine_windows(sums, curve="grumpkin")
    assert got.to_affine_bytes() == cg.to_affine64(cg.horner(sums))
    # ... and for the Pasta curves (255-bit moduli in the host's 4 x 64-bit arithmetic)
    for name in ("pallas", "vesta"):
        cx = __import__("importlib").import_module("oracle.cpu_" + name)
        sums_x = cx.g1_scalar_mul(cx.sample_points(5, 16), cx.sample_scalars(6, 16))
        assert m.MsmContext.combine_windows(sums_x, curve=name).to_affine_bytes() == cx.to_affine64(cx.horner(sums_x)), name
    # ... and for BLS12-381 (6 x 64-bit limbs on the host, 144-byte Jacobian records)
    from oracle import cpu_bls12_381 as cb

    sums_b = cb.g1_scalar_mul(cb.sample_points(7, 16), cb.sample_scalars(8, 16))
    assert len(sums_b) == 16 * 144
    got_b = m.MsmContext.combine_windows(sums_b, curve="bls12_381")
    assert got_b.to_affine_bytes() == cb.to_affine64(cb.horner(sums_b)) and len(got_b.to_affine_bytes()) == 96
    assert [g.to_affine_bytes() for g in m.MsmContext.combine_windows_batch(sums_b[: 8 * 144] + sums_b[: 8 * 144], 8, curve="bls12_381")] == [cb.to_affine64(cb.horner(sums_b[: 8 * 144]))] * 2
    # ... and for G2 of BN254 / BLS12-381 (Fq2 on the host's prime field, 192- / 288-byte Jacobian records) against the Python models
    import importlib

    for name in ("bn254_g2", "bls12_381_g2"):
        g2 = importlib.import_module("oracle." + name + "_ref")
        jac = []
        for k in range(16):  # 16 Jacobian points with z != 1 (and one identity)
            z = (g2.sample_scalar(11, k) % g2.P, g2.sample_scalar(12, k) % g2.P)
            pt = g2.mul(g2.sample_scalar(13, k), g2.G)
            z2 = g2.f2_sqr(z)
            jac.append((g2.f2_mul(pt[0], z2), g2.f2_mul(pt[1], g2.f2_mul(z2, z)), z) if k != 5 else ((0, 0), (0, 0), (0, 0)))
        sums_g = b"".join(g2.f2_to_bytes(x) + g2.f2_to_bytes(y) + g2.f2_to_bytes(z) for x, y, z in jac)
        want = None
        for x, y, z in reversed(jac):
            for _ in range(16):
                want = g2.add(want, want)
            want = g2.add(want, g2.j_to_affine((x, y, z)))
        got_g = m.MsmContext.combine_windows(sums_g, curve=name)
        assert len(got_g.xyz) == 3 * g2.CB and got_g.to_affine_bytes() == g2.affine_to_bytes(want), name
        assert g2.jacobian_bytes_to_affine(got_g.xyz) == want
        # ... and the library's own Jacobian -> affine conversion (an inversion in Fq2 on the host)
        aff = C.create_string_buffer(2 * g2.CB)
        assert L.msm_hip_g1_to_affine_curve({"bn254_g2": 5, "bls12_381_g2": 6}[name], got_g.xyz, aff) == 0
        assert aff.raw == g2.affine_to_bytes(want)
        assert L.msm_hip_g1_to_affine_curve({"bn254_g2": 5, "bls12_381_g2": 6}[name], bytes(3 * g2.CB), aff) == 1 and aff.raw == bytes(2 * g2.CB)  # identity
    out = C.create_string_buffer(288)
    assert L.msm_hip_combine_windows_curve(7, sums, 16, out) == -2  # unknown curve
    h = C.c_void_p()
    assert L.msm_hip_ctx_create_curve(C.byref(h), 0, 9) == -2


def test_wide_table_shapes_follow_the_scalar_field(built):
    """Host-only (msm_hip_wide_config): the wide fixed-base tables' digit width, table count, virtual windows and top-digit shift against an
    independent computation from every curve's scalar-field modulus: the top digit of a C-bit signed recode is at most
    ((r - 1 + bias) >> P) - 2^(C-1), P = C (T - 1), bias = the recode's constant; it must not pass 2^(C-1) (else the width cannot hold the
    curve's scalars -- 17 bits on BLS12-381); the library's bound on it (a 64-bit fraction, msm_hip.hip: wide_top_max) must never be below the exact
    value computed here.  The top digit is used unshifted since round 5 (interleaved virtual windows)."""
    import importlib

    import msm_webgpu_amd as m

    L = m.lib()
    moduli = {}
    for cid, name in ((0, "bn254_ref"), (1, "grumpkin_ref"), (2, "pallas_ref"), (3, "vesta_ref"), (4, "bls12_381_ref"), (5, "bn254_g2_ref"), (6, "bls12_381_g2_ref")):
        moduli[cid] = importlib.import_module("oracle." + name).R
    out = [C.c_int() for _ in range(4)]
    refs = [C.byref(x) for x in out]
    for cid, r in moduli.items():
        for bits in (16, 17, 18, 19, 20):
            t = (254 + bits) // bits
            pos, half = bits * (t - 1), 1 << (bits - 1)
            bias = sum(1 << (bits * w + bits - 1) for w in range(t))
            dmax = ((r - 1 + bias) >> pos) - half
            rc = L.msm_hip_wide_config(cid, bits, 1 << 20, *refs)
            if dmax > half:   # (the top digit is a bucket magnitude: 2^(C-1) itself is one)
                assert rc == -2, (cid, bits)
                continue
            # (round 5: interleaved virtual windows spread the narrow top digit by themselves: it is used unshifted)
            assert (rc, [x.value for x in out]) == (0, [bits, t, 1 << (bits - 16), 0]), (cid, bits)
        # the policy: 16 bits up to 2^16 bases, 17 up to 2^20 where 15 digits of 17 bits hold the scalars (else 19), 20 bits beyond
        fits17 = ((r - 1 + sum(1 << (17 * w + 16) for w in range(15))) >> 238) - (1 << 16) <= 1 << 16
        assert fits17 == (r < (1 << 254) + (1 << 200)), cid   # BN254, Grumpkin; Pallas and Vesta (2^254 + a 126-bit number); not BLS12-381
        for n, want in ((1000, 16), (1 << 16, 16), ((1 << 16) + 1, 17 if fits17 else 19), (1 << 20, 17 if fits17 else 19), ((1 << 20) + 1, 20), (1 << 24, 20)):
            assert L.msm_hip_wide_config(cid, 0, n, *refs) == 0 and out[0].value == want, (cid, n)
    assert L.msm_hip_wide_config(7, 0, 1, *refs) == -2 and L.msm_hip_wide_config(0, 15, 1, *refs) == -2


def test_header_is_plain_c_and_links_from_c(built, tmp_path):
    # the boundary is a C ABI: include/msm_hip.h compiles as strict C99 and a C program links against the library (what a cgo /
    # Rust FFI binding relies on); host-only entry points run without a GPU
    import subprocess

    import msm_webgpu_amd as m

    src = tmp_path / "c_caller.c"
    src.write_text(r'''
#include <stdio.h>
#include <string.h>
#include "msm_hip.h"
int main(void) {
  int nw = 0, nb = 0, b = 0, e = 0;
  unsigned char sums[96 * 16], out[96];
  if (msm_hip_abi_version() != 7) return 1;
  if (msm_hip_window_config(16, &nw, &nb) != MSM_HIP_OK || nw != 16 || nb != 32768) return 2;
  if (msm_hip_endomorphism_window_count(16) != 8) return 3;
  if (msm_hip_window_range(3, 8, 16, &b, &e) != MSM_HIP_OK || b != 6 || e != 8) return 4;
  memset(sums, 0, sizeof sums);  /* 16 identity window sums combine to the identity */
  if (msm_hip_combine_windows_bn254(sums, 16, out) != MSM_HIP_OK || out[64] != 0) return 5;
  if (strcmp(msm_hip_strerror(MSM_HIP_ERR_NO_BASES), "bases not set") != 0) return 6;
  msm_hip_oneshot_release();  /* nothing kept: a no-op */
  puts("c caller ok");
  return 0;
}
''')
    exe = str(tmp_path / "c_caller")
    so_dir = os.path.dirname(m.build())
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), str(src), "-o", exe,
                           "-L", so_dir, "-lmsm_hip", "-Wl,-rpath," + so_dir])
    assert "c caller ok" in subprocess.run([exe], capture_output=True, text=True, check=True).stdout
