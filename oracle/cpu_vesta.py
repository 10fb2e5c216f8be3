"""TEST INFRASTRUCTURE -- NOT PRODUCT CODE.  ctypes binding of oracle/liboracle_vesta.so: oracle/cpu.py's functions over the
Vesta build of the C restatement (oracle/bn254.c with -DORACLE_VESTA)."""
import importlib.util
import os

_here = os.path.dirname(os.path.abspath(__file__))
_spec = importlib.util.spec_from_file_location("oracle._cpu_vesta", os.path.join(_here, "cpu.py"))
_m = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_m)
_m._SO = os.path.join(_here, "liboracle_vesta.so")
globals().update({k: getattr(_m, k) for k in dir(_m) if not k.startswith("__")})
