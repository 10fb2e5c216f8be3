"""The C++ host-side mirror of the reference API (include/msm_hip.hpp) compiles against the C ABI, fails loudly without a
GPU, and -- on a GPU -- reproduces the oracle's result bit-exactly (the test reads like src/lib.rs:152-167)."""
import os
import subprocess

import pytest
import torch

from oracle import cpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def exe(built, tmp_path_factory):
    out = str(tmp_path_factory.mktemp("cpp") / "test_host_api")
    libdir = os.path.join(ROOT, "msm-webgpu_amd")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "test_host_api.cpp"),
                           "-L", libdir, "-lmsm_hip", "-Wl,-rpath," + libdir, "-o", out])
    return out


def test_cpp_api_fails_loudly_without_gpu(exe):
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    r = subprocess.run([exe, "--no-device"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.gpu
def test_cpp_api_matches_oracle(exe, tmp_path):
    n = 5000
    points, scalars = cpu.sample_points(201, n), cpu.sample_scalars(202, n)
    want = cpu.to_affine64(cpu.cpu_msm(points, scalars))
    for name, data in (("p.bin", points), ("s.bin", scalars), ("w.bin", want)):
        (tmp_path / name).write_bytes(data)
    r = subprocess.run([exe, str(tmp_path / "p.bin"), str(tmp_path / "s.bin"), str(tmp_path / "w.bin")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "host api ok" in r.stdout
