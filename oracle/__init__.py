"""TEST INFRASTRUCTURE -- NOT PRODUCT CODE.

CPU oracle for the BN254 G1 MSM hot path: `oracle.cpu` (ctypes over the plain-C restatement
`oracle/bn254.c`) and `oracle.bn254_ref` (pure-Python big-int model).  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this package; the product
(`msm-webgpu_amd/`) never does.  Parity status: see `oracle/bn254.h`.
"""
