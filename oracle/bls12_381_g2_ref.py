"""TEST INFRASTRUCTURE -- NOT PRODUCT CODE.  The G2 model of oracle/bn254_g2_ref.py instantiated for BLS12-381 (SURVEY.md section 8f-4
"other curves / G2"): the twist y^2 = x^3 + 4 (1 + u) over Fq2 = Fq[u] / (u^2 + 1) of the 381-bit p, scalars modulo the 255-bit r; an Fq2
element is 2 x 48 bytes on the wire.  A private copy of the module with P, R, B, G, FB, CB and its G1 model rebound (its functions read
them as globals).  Pinned in tests/test_oracle_g2.py: the standard generator lies on the twist and has order r."""
import importlib.util
import os
import sys

from . import bls12_381_ref as _bls

_spec = importlib.util.spec_from_file_location("oracle._bls12_381_g2_model", os.path.join(os.path.dirname(os.path.abspath(__file__)), "bn254_g2_ref.py"),
                                               submodule_search_locations=None)
_m = importlib.util.module_from_spec(_spec)
_m.__package__ = __package__
_spec.loader.exec_module(_m)
_m._g1 = _bls
_m.P, _m.R = _bls.P, _bls.R
_m.FB, _m.CB = 48, 96
_m.B = (4, 4)
# the standard generator of the order-r subgroup of the twist (public parameter of the curve: IETF pairing-friendly-curves draft, zkcrypto/bls12_381)
_m.G = ((0x024aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb8,
         0x13e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049334cf11213945d57e5ac7d055d042b7e),
        (0x0ce5d527727d6e118cc9cdc6da2e351aadfd9baa8cbdd3a76d429a695160d12c923ac9cc3baca289e193548608b82801,
         0x0606c4a02ea734cc32acd2b02bc28b99cb3e287e85a763af267492ab572e99ab3f370d275cec1da1aaa9075ff05f79be))
assert _m.is_on_curve(_m.G)
globals().update({k: getattr(_m, k) for k in dir(_m) if not k.startswith("__")})
sys.modules[__name__].__dict__["_model"] = _m
