#!/usr/bin/env python3
"""Third code-generation gate (round 4): LLVM's own machine verifier over the device code of every translation unit.

Background (profiles/r04_asm_everywhere_rootcause.txt).  Round 3's diagnostic build with the inline-assembly multipliers in
every kernel (since round 4, with -fno-slp-vectorize, the product's shape) returned wrong bucket sums in the 14-limb (BLS12-381) and Fq2 (BN254 G2) units -- the stitch's P + P path.  Cause: a
miscompile, not the assembly.  clang's SLP vectoriser packs the limb arrays of the loop-carried accumulator into <2 x i32> values, the
AMDGPU backend keeps those as 64 / 128-bit register tuples, and LLVM's register coalescer, joining the copy of the doubling's last y limb
(a V_ADD3_U32) into such a tuple, marks that definition `dead` although its lane is live out of the block.  The register allocator then
lets the join's tuple copy overwrite it: Y's top limb is garbage after a doubling.  `llc -verify-machineinstrs` reports it right after the
coalescer ("Live range continues after dead def flag") -- in exactly the two kernels that failed on the GPU, and nowhere in the product.

The check: every translation unit is compiled once more, device side only, with the build's own flags plus `-mllvm
-verify-machineinstrs`; a clean unit compiles, a miscompiled one aborts naming function and instruction.  Results are cached under
build/ (key: the sources' and flags' hash) -- the verifier makes a unit's compile 2 - 3 x slower.

usage: check_machine_verifier.py            every unit with the flags of the current environment (MSM_HIP_SLP=1: the miscompiled variant; MSM_HIP_ASM_SMVP_ONLY, MSM_HIP_EXTRA_FLAGS ...)
exit status 0 = clean, 1 = the verifier fired.
"""
import hashlib
import json
import os
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "msm-webgpu_amd"))
import build as _b  # noqa: E402

CACHE = os.path.join(ROOT, "build", "machine_verifier_cache.json")


def sources_hash(flags):
    h = hashlib.sha256(" ".join(flags).encode())
    for name in sorted(_b.SOURCES):
        path = os.path.join(_b.CSRC, name)
        if os.path.exists(path):
            h.update(name.encode())
            h.update(open(path, "rb").read())
    return h.hexdigest()


def verify_unit(unit, flags):
    """-> list of (function, message) the machine verifier reported for `unit` (empty: clean)"""
    with tempfile.TemporaryDirectory(prefix="msm_hip_verify_") as tmp:
        cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + flags + ["--cuda-device-only", "-mllvm", "-verify-machineinstrs", "-c",
                                                                          os.path.join(_b.CSRC, unit), "-o", os.path.join(tmp, "unit.o")]
        p = subprocess.run(cmd, cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True, errors="replace")
    found, msg = [], None
    for ln in p.stdout.split("\n"):
        m = re.match(r"\*\*\* Bad machine code: (.*) \*\*\*", ln)
        if m:
            msg = m.group(1)
        m = re.match(r"- function:\s+(\S+)", ln)
        if m and msg:
            found.append((m.group(1), msg))
            msg = None
    if p.returncode != 0 and not found:  # the compile failed for another reason: not a verdict
        raise RuntimeError("%s: hipcc failed without a machine-verifier report:\n%s" % (unit, p.stdout[-2000:]))
    return found


def check(units=None, use_cache=True):
    """-> {unit: [(function, message), ...]} for every translation unit of the build, with the current environment's flags"""
    units = units or _b.TRANSLATION_UNITS
    flags = _b.compile_flags()
    key = sources_hash(flags)
    cache = {}
    if use_cache and os.path.exists(CACHE):
        try:
            cache = json.load(open(CACHE))
        except ValueError:
            cache = {}
    todo = [u for u in units if key + ":" + u not in cache]
    with ThreadPoolExecutor(max_workers=min(len(todo) or 1, os.cpu_count() or 1)) as pool:
        for u, found in zip(todo, pool.map(lambda u: verify_unit(u, flags), todo)):
            cache[key + ":" + u] = found
    if todo and use_cache:
        os.makedirs(os.path.dirname(CACHE), exist_ok=True)
        json.dump(cache, open(CACHE, "w"))
    return {u: [tuple(f) for f in cache[key + ":" + u]] for u in units}


def main():
    bad = 0
    for unit, found in check(use_cache="--no-cache" not in sys.argv).items():
        print("%s: %s" % (unit, "clean" if not found else "%d machine-verifier reports" % len(found)))
        for fn, msg in found:
            print("  MISCOMPILE in %s: %s" % (fn, msg))
            bad += 1
    return 1 if bad else 0


if __name__ == "__main__":
    raise SystemExit(main())
