"""Build libmsm_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# MSM_HIP_SO: load (and build into) another file, e.g. a diagnostic variant next to the product library
SO = os.environ.get("MSM_HIP_SO") or os.path.join(HERE, "libmsm_hip.so")
SOURCES = ["msm_hip.hip", "msm_kernels.h", "g1.h", "fq29.h", "fq29_asm.h", "host_g1.h", "glv.h", "msm_mgpu.h", "curve_select.h", "curve_unit.h", "grumpkin_constants.h", "bn254_constants.h", "pallas_constants.h", "vesta_constants.h"]
HEADER = os.path.join(HERE, "..", "include", "msm_hip.h")
TEMPS = os.path.join(HERE, "..", "build", "temps" if not os.environ.get("MSM_HIP_SO") else "temps_" + os.path.basename(SO))
DEVICE_ASM = os.path.join(TEMPS, "msm_hip-hip-amdgcn-amd-amdhsa-gfx950.s")  # written by build()


def device_asm_is_current():
    """True if build() left the device assembly of the CURRENT sources behind (same staleness rule as the library itself)."""
    if needs_build() or not os.path.exists(DEVICE_ASM):
        return False
    return os.path.getmtime(DEVICE_ASM) >= max(os.path.getmtime(os.path.join(CSRC, f)) for f in SOURCES if os.path.exists(os.path.join(CSRC, f)))


def needs_build():
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [HEADER]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 -shared -fPIC csrc/msm_hip.hip -> msm-webgpu_amd/libmsm_hip.so"""
    if not force and not needs_build():
        return SO
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    # the compiler's intermediate files are kept (build/temps/, git-ignored): the device assembly among them is what the code-generation
    # gate reads (tools/check_long_branch_hazard.py, tests/test_codegen_hazards.py) instead of compiling everything a second time
    os.makedirs(TEMPS, exist_ok=True)
    cmd = [hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared", "-save-temps=cwd",
           os.path.join(CSRC, "msm_hip.hip"), "-o", SO + ".tmp"]
    if os.environ.get("MSM_HIP_NO_ASM") == "1":  # the C++ multipliers everywhere (the inline-assembly ones are only in g1_madd)
        cmd.insert(1, "-DFQ29_NO_ASM")
    if os.environ.get("MSM_HIP_ASM_EVERYWHERE") == "1":  # diagnostic: the inline-assembly multipliers in every kernel
        cmd.insert(1, "-DFQ29_ASM_EVERYWHERE")
    cmd[1:1] = os.environ.get("MSM_HIP_EXTRA_FLAGS", "").split()
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=TEMPS)
    os.replace(SO + ".tmp", SO)
    return SO


if __name__ == "__main__":
    print(build(force=True, verbose=True))
