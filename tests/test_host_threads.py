"""Sanitizers on the CPU build of the threaded host code: csrc/host_worker.h (the per-device worker thread of
msm_hip_mgpu_*) and csrc/host_pool.h (the combine pool), hammered by tests/host_harness/threads_harness.cpp under ThreadSanitizer and under
AddressSanitizer + UBSan, with the window combines checked against the ORACLE's Horner (oracle/bn254.c).  No GPU (and no GPU sanitizer: this
pool has none)."""
import os
import struct
import subprocess

import pytest

from oracle import bn254_ref as ref
from oracle import cpu
from tests.util import jacobian_bytes, rng

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "host_harness", "threads_harness.cpp")
INC = os.path.join(ROOT, "msm-webgpu_amd", "csrc")


@pytest.fixture(scope="module")
def vectors(tmp_path_factory):
    """window sums (random multiples of the generator as Jacobian records, some identities) and the oracle's Horner of each case"""
    r = rng(77)
    cases, windows = 12, 16
    blob = struct.pack("<II", cases, windows)
    for c in range(cases):
        sums = b"".join(jacobian_bytes(None if (c + w) % 5 == 0 else ref.mul(r.randrange(1, ref.R), ref.G), r) for w in range(windows))
        blob += sums + cpu.horner(sums, 16)
    path = tmp_path_factory.mktemp("threads") / "vectors.bin"
    path.write_bytes(blob)
    return str(path)


def _build(tmp, name, san, extra=()):
    exe = str(tmp / name)
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-pthread", "-I", INC] + list(san) + list(extra) + [SRC, "-o", exe])
    return exe


def _run(exe, vectors, rounds="3"):
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1 exitcode=66", ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="halt_on_error=1")
    return subprocess.run([exe, vectors, rounds], env=env, capture_output=True, text=True, timeout=600)


def test_threaded_host_code_under_thread_sanitizer(tmp_path, vectors):
    r = _run(_build(tmp_path, "tsan", ["-fsanitize=thread"]), vectors)
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stderr[-3000:]


def test_threaded_host_code_under_address_and_ub_sanitizers(tmp_path, vectors):
    r = _run(_build(tmp_path, "asan", ["-fsanitize=address,undefined", "-fno-sanitize-recover=all"]), vectors)
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stderr[-3000:]


def test_thread_sanitizer_run_is_not_vacuous(tmp_path, vectors):
    """the same harness with the worker's queue push left unlocked must FAIL under ThreadSanitizer"""
    r = _run(_build(tmp_path, "tsan_broken", ["-fsanitize=thread"], ["-DHARNESS_BREAK_WORKER_LOCK"]), vectors, "1")
    assert r.returncode != 0 and "ThreadSanitizer" in r.stderr, (r.returncode, r.stderr[-1000:])
