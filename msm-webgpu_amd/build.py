"""Build libmsm_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# MSM_HIP_SO: load (and build into) another file, e.g. a diagnostic variant next to the product library
SO = os.environ.get("MSM_HIP_SO") or os.path.join(HERE, "libmsm_hip.so")
SOURCES = ["msm_hip.hip", "curve_grumpkin.hip", "curve_pallas.hip", "curve_vesta.hip", "curve_bls12_381.hip", "curve_bn254_g2.hip", "curve_bls12_381_g2.hip", "fq2.h", "bn254_g2_constants.h", "bls12_381_g2_constants.h", "fq28x14_asm.h", "bls12_381_constants.h", "curve_ops.h", "msm_kernels.h", "g1.h", "fq29.h", "fq29_asm.h", "host_g1.h", "host_pool.h", "host_worker.h", "glv.h", "msm_mgpu.h", "curve_select.h", "curve_unit.h", "grumpkin_constants.h", "bn254_constants.h", "pallas_constants.h", "vesta_constants.h"]
HEADER = os.path.join(HERE, "..", "include", "msm_hip.h")
TEMPS = os.path.join(HERE, "..", "build", "temps" if not os.environ.get("MSM_HIP_SO") else "temps_" + os.path.basename(SO))


def device_asm_files():
    """The device assembly of every translation unit, as build() leaves it behind."""
    return [os.path.join(TEMPS, os.path.splitext(u)[0] + "-hip-amdgcn-amd-amdhsa-gfx950.s") for u in TRANSLATION_UNITS]


def device_asm_is_current():
    """True if build() left the device assembly of the CURRENT sources behind (same staleness rule as the library itself)."""
    if needs_build() or not all(os.path.exists(f) for f in device_asm_files()):
        return False
    newest = max(os.path.getmtime(os.path.join(CSRC, f)) for f in SOURCES if os.path.exists(os.path.join(CSRC, f)))
    return all(os.path.getmtime(f) >= newest for f in device_asm_files())


def build_stamp():
    """what the library on disk must have been built FROM to be the product: a hash over the compile flags (the environment switches of the
    diagnostic builds included -- MSM_HIP_SLP, MSM_HIP_NO_ASM, MSM_HIP_EXTRA_FLAGS ...) and the contents of every source.  Written next to the
    library by build(); a library whose stamp differs (built once with variant flags, or from other sources) is rebuilt instead of being
    silently reused, and the code-generation gates (tools/check_machine_verifier.py, tools/check_long_branch_hazard.py) check the stamp of the
    library they vouch for."""
    import hashlib

    h = hashlib.sha256()
    h.update("\0".join(compile_flags()).encode())
    for f in sorted(SOURCES) + [HEADER]:
        path = f if os.path.isabs(f) else os.path.join(CSRC, f)
        if os.path.exists(path):
            h.update(b"\0" + os.path.basename(path).encode() + b"\0")
            with open(path, "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()


def stamp_path():
    return SO + ".stamp"


def stamp_is_current():
    """True if the library on disk was built from the current sources with the current flags"""
    try:
        with open(stamp_path()) as f:
            return f.read().strip() == build_stamp()
    except OSError:
        return False


def variant_flags():
    """the environment switches that change the generated code (the diagnostic builds)"""
    return [k for k in ("MSM_HIP_SLP", "MSM_HIP_NO_ASM", "MSM_HIP_ASM_SMVP_ONLY", "MSM_HIP_EXTRA_FLAGS") if os.environ.get(k)]


def needs_build():
    if not os.path.exists(SO):
        return True
    if os.environ.get("MSM_HIP_SO"):
        # a diagnostic library named explicitly (built here with variant flags, run on the GPU box without them in the environment): its own
        # modification time decides -- the stamp rule below is the PRODUCT's
        t = os.path.getmtime(SO)
        return any(os.path.getmtime(d) > t for d in [os.path.join(CSRC, s) for s in SOURCES] + [HEADER] if os.path.exists(d))
    if os.path.exists(stamp_path()):
        return not stamp_is_current()
    # a library without a stamp (built by an older checkout): the modification times decide, as they used to
    t = os.path.getmtime(SO)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [HEADER]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


TRANSLATION_UNITS = ["msm_hip.hip", "curve_bls12_381_g2.hip", "curve_bn254_g2.hip", "curve_bls12_381.hip", "curve_grumpkin.hip", "curve_pallas.hip", "curve_vesta.hip"]  # host + BN254's unit; one per further curve


def compile_flags():
    """hipcc flags of every translation unit (the environment switches of the diagnostic builds included)"""
    # -fno-slp-vectorize: the SLP vectoriser packs the limb arrays into <2 x i32> values, which the backend keeps in 64 / 128-bit register
    # tuples -- the shape on which this LLVM's register coalescer miscompiled the all-assembly diagnostic build (profiles/
    # r04_asm_everywhere_rootcause.txt; gate: tools/check_machine_verifier.py).  Without it: same registers, 1 - 2 % shorter single-MSM
    # latency, throughput unchanged (profiles/r04_ab_noslp.txt).  MSM_HIP_SLP=1 restores the compiler's default.
    flags = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC"]
    if os.environ.get("MSM_HIP_SLP") != "1":
        flags.append("-fno-slp-vectorize")
    if os.environ.get("MSM_HIP_NO_ASM") == "1":  # the C++ multipliers everywhere
        flags.append("-DFQ29_NO_ASM")
    if os.environ.get("MSM_HIP_ASM_SMVP_ONLY") == "1":  # rounds 1 - 3: the inline-assembly multipliers only in the SMVP's mixed addition, C++ elsewhere
        flags.append("-DFQ29_ASM_SMVP_ONLY")
    return flags + os.environ.get("MSM_HIP_EXTRA_FLAGS", "").split()


def build(force=False, verbose=False):
    """hipcc --offload-arch=gfx950: every translation unit of csrc/ to an object (in parallel), then -shared -> msm-webgpu_amd/libmsm_hip.so"""
    if not force and not needs_build():
        return SO
    if variant_flags() and not os.environ.get("MSM_HIP_SO"):
        # the product library is only ever built with the gated flags: a variant build must name its own file
        raise RuntimeError("variant build flags (%s) need MSM_HIP_SO=<another file>: libmsm_hip.so is the product" % ", ".join(variant_flags()))
    from concurrent.futures import ThreadPoolExecutor

    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    # the compiler's intermediate files are kept (build/temps/, git-ignored): the device assembly among them is what the code-generation
    # gate reads (tools/check_long_branch_hazard.py, tests/test_codegen_hazards.py) instead of compiling everything a second time
    os.makedirs(TEMPS, exist_ok=True)
    flags = compile_flags()

    def compile_unit(name):
        obj = os.path.join(TEMPS, os.path.splitext(name)[0] + ".o")
        cmd = [hipcc] + flags + ["-save-temps=cwd", "-c", os.path.join(CSRC, name), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd, cwd=TEMPS)
        return obj

    with ThreadPoolExecutor(max_workers=min(len(TRANSLATION_UNITS), os.cpu_count() or 1)) as pool:
        objs = list(pool.map(compile_unit, TRANSLATION_UNITS))
    cmd = [hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", "--hip-link"] + objs + ["-o", SO + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=TEMPS)
    os.replace(SO + ".tmp", SO)
    with open(stamp_path(), "w") as f:
        f.write(build_stamp() + "\n")
    return SO


if __name__ == "__main__":
    print(build(force=True, verbose=True))
