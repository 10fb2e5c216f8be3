"""TEST INFRASTRUCTURE -- NOT PRODUCT CODE.  ctypes binding of oracle/liboracle_bls12_381_g2.so: oracle/cpu.py's functions over the G2 build of
the C restatement (oracle/bn254.c with -DORACLE_G2 -DORACLE_BLS12_381: coordinates in Fq2, c0 || c1 on the wire)."""
import importlib.util
import os

_here = os.path.dirname(os.path.abspath(__file__))
_spec = importlib.util.spec_from_file_location("oracle._cpu_bls12_381_g2", os.path.join(_here, "cpu.py"))
_m = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_m)
_m._SO = os.path.join(_here, "liboracle_bls12_381_g2.so")
globals().update({k: getattr(_m, k) for k in dir(_m) if not k.startswith("__")})
