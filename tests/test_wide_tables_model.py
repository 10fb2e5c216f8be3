"""The wide fixed-base tables' arithmetic as a model over Z_r (CPU only; test infrastructure).

The MSM is linear in the points, so the additive group Z_r stands in for the curve: "point" i is an integer g_i, "sum_i s_i P_i" is
sum_i s_i g_i mod r.  The model follows csrc/msm_kernels.h (k_count_wide / wide_digit: signed C-bit digits by one biased addition, the top
digit shifted against a top table that is `shift` doublings short -- 0 since round 5 --, magnitude m -> virtual window (m - 1) mod V and the slot of
value (m - 1) / V + 1, slot 0 carrying 2^15) and csrc/host_g1.h (combine_wide: V sum_vw W_vw - sum_vw (V - 1 - vw) TC_vw), with the shape -- digit width, tables, virtual
windows, top shift -- taken from the library's own host-only helper (msm_hip_wide_config), for every curve's scalar field and every width."""
import ctypes as C
import importlib
import random

import pytest

CURVES = ((0, "bn254_ref"), (1, "grumpkin_ref"), (2, "pallas_ref"), (3, "vesta_ref"), (4, "bls12_381_ref"))


def wide_msm_model(r, scalars, points, bits, tables, vwin, shift):
    half = 1 << (bits - 1)
    pos_top = bits * (tables - 1)
    table = [[(g << (bits * w)) % r for g in points] for w in range(tables - 1)]
    table.append([(g << (pos_top - shift)) % r for g in points])                      # the top table: `shift` doublings short
    bias = sum(1 << (bits * w + bits - 1) for w in range(tables))
    buckets = [[0] * (1 << 15) for _ in range(vwin)]
    for i, s in enumerate(scalars):
        t = s + bias
        for w in range(tables):
            if w < tables - 1:
                b = (t >> (bits * w)) & ((1 << bits) - 1)
                mag, neg = (b - half, False) if b >= half else (half - b, True)
            else:
                d = (t >> pos_top) - half                                             # the top digit with everything above it: never negative
                assert 0 <= d and (d << shift) <= half, "a scalar below r must fit"
                mag, neg = d << shift, False
            if mag == 0:
                continue
            hi, slot = (mag - 1) % vwin, ((mag - 1) // vwin + 1) & 0x7FFF               # msm_kernels.h: wide_key / wide_slot (interleaved)
            assert hi < vwin
            buckets[hi][slot] = (buckets[hi][slot] + (-table[w][i] if neg else table[w][i])) % r
    total, run, minus = 0, 0, 0
    for vw in range(vwin):                                                            # host_g1.h: combine_wide_strided
        w_vw = sum((slot if slot else 1 << 15) * v for slot, v in enumerate(buckets[vw])) % r
        tc_vw = sum(buckets[vw]) % r
        total = (total + w_vw) % r
        if vw < vwin - 1:
            run = (run + tc_vw) % r
            minus = (minus + run) % r
    return (total * vwin - minus) % r


@pytest.mark.parametrize("cid,name", CURVES)
def test_wide_tables_model_reproduces_the_msm(built, cid, name):
    import msm_webgpu_amd as m

    r = importlib.import_module("oracle." + name).R
    out = [C.c_int() for _ in range(4)]
    rnd = random.Random(7000 + cid)
    n = 64
    points = [rnd.randrange(1, r) for _ in range(n)]
    edge = [0, 1, 2, r - 1, r - 2, (1 << 15), (1 << 15) + 1, (1 << 16) - 1, (1 << 16), r >> 1, (r >> 1) + 1]
    for bits in (16, 17, 18, 19, 20):
        if m.lib().msm_hip_wide_config(cid, bits, n, *[C.byref(x) for x in out]) != 0:
            assert bits == 17 and r.bit_length() == 255 and r > (1 << 254) + (1 << 200)   # only BLS12-381's field does not fit 15 x 17 bits
            continue
        got_bits, tables, vwin, shift = (x.value for x in out)
        assert got_bits == bits and tables == (254 + bits) // bits and vwin == 1 << (bits - 16)
        scalars = edge + [rnd.randrange(r) for _ in range(n - len(edge))]
        # digits at the seams of the virtual windows in every position
        for w in range(tables - 1):
            for d in ((1 << 15) - 1, 1 << 15, (1 << 15) + 1, (1 << (bits - 1)) - 1, 1 << (bits - 1), (1 << bits) - 1):
                v = d << (bits * w)
                if v < r:
                    scalars[rnd.randrange(len(edge), n)] = v
        want = sum(s * g for s, g in zip(scalars, points)) % r
        assert wide_msm_model(r, scalars, points, bits, tables, vwin, shift) == want, (name, bits)
