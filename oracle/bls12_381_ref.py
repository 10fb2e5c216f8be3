"""TEST INFRASTRUCTURE -- NOT PRODUCT CODE.  The pure-Python big-integer model of oracle/bn254_ref.py instantiated for BLS12-381 G1
(SURVEY.md section 8f-4 "other curves"; reference README.md: "Implement cuzk on other curves"): y^2 = x^3 + 4 over the 381-bit p, scalars
modulo the 255-bit r, 48-byte coordinates on the wire.  A private copy of the module with P, R, B, G, CB rebound (its functions read them
as globals)."""
import importlib.util
import os
import sys

BLS_P = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
BLS_R = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
_spec = importlib.util.spec_from_file_location("oracle._bls12_381_model", os.path.join(os.path.dirname(os.path.abspath(__file__)), "bn254_ref.py"))
_m = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_m)
_m.P, _m.R = (BLS_P, BLS_R)
_m.B = 4
_m.CB = 48
# the standard generator of the order-r subgroup (public parameter of the curve)
_m.G = (0x17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb,
        0x08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1)
assert _m.is_on_curve(_m.G)
globals().update({k: getattr(_m, k) for k in dir(_m) if not k.startswith("__")})
sys.modules[__name__].__dict__["_model"] = _m
