"""msm-webgpu_amd -- MI355X-native BN254 G1 multi-scalar multiplication (cuZK-style Pippenger in HIP).

Host-side mirror of the reference's public surface for this path (/root/reference/src/lib.rs:19-82):
`run_webgpu_msm`, `compute_msm`, `points_to_bytes`, `scalars_to_bytes`, `sample_points`, `sample_scalars`,
plus the persistent `MsmContext` the C ABI (include/msm_hip.h) adds.  All arithmetic runs in
libmsm_hip.so (hand-written HIP for gfx950); there is no CPU fallback -- using the API without the
built library, or without a GPU, raises.
"""
from .api import (  # noqa: F401
    G1,
    MsmContext,
    MsmHipError,
    MultiGpuMsm,
    compute_msm,
    lib,
    points_to_bytes,
    run_webgpu_msm,
    sample_points,
    sample_scalars,
    scalars_to_bytes,
)
from .build import build  # noqa: F401

__all__ = ["G1", "MsmContext", "MultiGpuMsm", "MsmHipError", "compute_msm", "run_webgpu_msm", "points_to_bytes", "scalars_to_bytes",
           "sample_points", "sample_scalars", "build", "lib"]
