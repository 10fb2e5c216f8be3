"""TEST INFRASTRUCTURE -- NOT PRODUCT CODE.  The pure-Python big-integer model of oracle/bn254_ref.py instantiated for Pallas (the
Pasta cycle, SURVEY.md section 8f-4 "other curves"; the reference keeps dead Pallas shaders under src/naive/wgsl/pallas):
y^2 = x^3 + 5, generator (-1, 2); Pallas lives over p with q points, Vesta over q with p points.  A private copy of the module
with P, R, B, G rebound (its functions read them as globals)."""
import importlib.util
import os
import sys

PALLAS_P = 0x40000000000000000000000000000000224698fc094cf91b992d30ed00000001
VESTA_P = 0x40000000000000000000000000000000224698fc0994a8dd8c46eb2100000001
_spec = importlib.util.spec_from_file_location("oracle._pallas_model", os.path.join(os.path.dirname(os.path.abspath(__file__)), "bn254_ref.py"))
_m = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_m)
_m.P, _m.R = (PALLAS_P, VESTA_P)
_m.B = 5
_m.G = (_m.P - 1, 2)
assert _m.is_on_curve(_m.G)
globals().update({k: getattr(_m, k) for k in dir(_m) if not k.startswith("__")})
sys.modules[__name__].__dict__["_model"] = _m
