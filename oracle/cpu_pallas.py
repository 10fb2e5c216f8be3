"""TEST INFRASTRUCTURE -- NOT PRODUCT CODE.  ctypes binding of oracle/liboracle_pallas.so: oracle/cpu.py's functions over the
Pallas build of the C restatement (oracle/bn254.c with -DORACLE_PALLAS)."""
import importlib.util
import os

_here = os.path.dirname(os.path.abspath(__file__))
_spec = importlib.util.spec_from_file_location("oracle._cpu_pallas", os.path.join(_here, "cpu.py"))
_m = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_m)
_m._SO = os.path.join(_here, "liboracle_pallas.so")
globals().update({k: getattr(_m, k) for k in dir(_m) if not k.startswith("__")})
