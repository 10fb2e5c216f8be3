"""TEST INFRASTRUCTURE -- NOT PRODUCT CODE.  The pure-Python big-integer model of oracle/bn254_ref.py instantiated for Grumpkin,
BN254's cycle partner (SURVEY.md section 8f-4 "other curves"): base field = BN254's scalar field r, scalar field = BN254's base
field p, y^2 = x^3 - 17.  Every function of bn254_ref is available here with these parameters (the module's functions read
P, R, B, G as globals; this is a private copy of the module with the four rebound)."""
import importlib.util
import os
import sys

_spec = importlib.util.spec_from_file_location("oracle._grumpkin_model", os.path.join(os.path.dirname(os.path.abspath(__file__)), "bn254_ref.py"))
_m = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_m)
_m.P, _m.R = _m.R, _m.P          # Grumpkin's base field is BN254's Fr, its scalar field BN254's Fq
_m.B = -17
_m.G = (1, _m.sqrt_mod(1 - 17))  # (1, sqrt(-16))
if _m.G[1] > _m.P - _m.G[1]:      # the smaller root, as the curve's usual generator
    _m.G = (1, _m.P - _m.G[1])
assert _m.is_on_curve(_m.G)
globals().update({k: getattr(_m, k) for k in dir(_m) if not k.startswith("__")})
sys.modules[__name__].__dict__["_model"] = _m
