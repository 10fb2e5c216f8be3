"""TEST INFRASTRUCTURE -- NOT PRODUCT CODE.  ctypes binding of oracle/liboracle_bn254.so (see oracle/bn254.h)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle_bn254.so")
_lib = None


def build(force=False):
    """Compile the C restatement with gcc (building the checker is not using it)."""
    src = [os.path.join(_HERE, f) for f in ("bn254.c", "bn254.h")]
    if force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        u8p, i32p, sz = C.c_char_p, C.POINTER(C.c_int32), C.c_size_t
        L.oracle_fq_op.argtypes = [C.c_int, u8p, u8p, u8p, sz]
        L.oracle_g1_op.argtypes = [C.c_int, u8p, u8p, u8p, sz]
        L.oracle_g1_scalar_mul.argtypes = [u8p, u8p, u8p, sz]
        L.oracle_g1_to_affine64.argtypes = [u8p, u8p]
        L.oracle_points_on_curve.argtypes = [u8p, sz]
        L.oracle_msm_bn254_g1.argtypes = [u8p, u8p, sz, u8p]
        L.oracle_msm_bn254_g1_mt.argtypes = [u8p, u8p, sz, C.c_int, u8p]
        L.oracle_decompose_scalars_signed.argtypes = [u8p, sz, C.c_int, C.c_int, C.c_void_p]
        L.oracle_transpose.argtypes = [C.c_void_p, sz, C.c_int, C.c_void_p, C.c_void_p]
        L.oracle_smvp_signed.argtypes = [C.c_void_p, C.c_void_p, u8p, sz, C.c_int, u8p]
        L.oracle_bucket_reduction.argtypes = [C.c_int, u8p, C.c_int, C.c_int, u8p]
        L.oracle_parallel_bucket_reduction_1.argtypes = [u8p, C.c_int, C.c_int, u8p, u8p]
        L.oracle_parallel_bucket_reduction_2.argtypes = [u8p, u8p, C.c_int, C.c_int, u8p]
        L.oracle_horner.argtypes = [u8p, C.c_int, C.c_int, u8p]
        L.oracle_msm_cuzk_model.argtypes = [u8p, u8p, sz, C.c_int, u8p]
        L.oracle_sample_scalars.argtypes = [C.c_uint64, sz, sz, u8p]
        L.oracle_sample_points.argtypes = [C.c_uint64, sz, sz, u8p]
        L.oracle_constants.argtypes = [u8p, u8p, u8p, u8p, C.POINTER(C.c_uint64)]
        L.oracle_coord_bytes.restype = C.c_int
        _lib = L
    return _lib


def _buf(n):
    return C.create_string_buffer(n)


def coord_bytes():
    """Bytes of a coordinate on this build's wire: 32, or 48 for the BLS12-381 build (points 2 x, Jacobian records 3 x that)."""
    return lib().oracle_coord_bytes()


def _cb():
    return coord_bytes()


def fq_op(op, a, b=None):
    """op in {add, sub, mul, sqr, neg, inv}; a, b: bytes of n x 32 B canonical LE."""
    code = {"add": 0, "sub": 1, "mul": 2, "sqr": 3, "neg": 4, "inv": 5}[op]
    n = len(a) // _cb()
    out = _buf(_cb() * n)
    lib().oracle_fq_op(code, a, b, out, n)
    return out.raw


def g1_op(op, a, b=None):
    code = {"add": 0, "double": 1, "negate": 2}[op]
    n = len(a) // (3 * _cb())
    out = _buf(3 * _cb() * n)
    lib().oracle_g1_op(code, a, b, out, n)
    return out.raw


def g1_scalar_mul(points_xy, scalars):
    n = len(points_xy) // (2 * _cb())
    out = _buf(3 * _cb() * n)
    lib().oracle_g1_scalar_mul(points_xy, scalars, out, n)
    return out.raw


def to_affine64(xyz):
    """96 B Jacobian -> 64 B canonical affine (64 zero bytes for the identity)."""
    out = _buf(2 * _cb())
    lib().oracle_g1_to_affine64(bytes(xyz), out)
    return out.raw


def points_on_curve(xy):
    return bool(lib().oracle_points_on_curve(xy, len(xy) // (2 * _cb())))


def cpu_msm(points_xy, scalars, n_threads=1):
    """≙ cpu_msm (src/lib.rs:45-47).  Returns the 96 B Jacobian result."""
    n = len(scalars) // 32
    assert len(points_xy) == 2 * _cb() * n
    out = _buf(3 * _cb())
    lib().oracle_msm_bn254_g1_mt(points_xy, scalars, n, n_threads, out)
    return out.raw


def decompose_scalars_signed(scalars, num_words=16, word_size=16):
    n = len(scalars) // 32
    digits = np.zeros((num_words, n), dtype=np.int32)
    rc = lib().oracle_decompose_scalars_signed(scalars, n, num_words, word_size, digits.ctypes.data)
    if rc != 0:
        raise ValueError("final carry is 1")
    return digits


def transpose(digits_w, num_columns):
    digits_w = np.ascontiguousarray(digits_w, dtype=np.int32)
    n = digits_w.shape[0]
    col_ptr = np.zeros(num_columns + 1, dtype=np.int32)
    val = np.zeros(max(n, 1), dtype=np.int32)
    lib().oracle_transpose(digits_w.ctypes.data, n, num_columns, col_ptr.ctypes.data, val.ctypes.data)
    return col_ptr, val[:n]


def smvp_signed(col_ptr, val_idxs, points_xy, num_columns):
    col_ptr = np.ascontiguousarray(col_ptr, dtype=np.int32)
    val_idxs = np.ascontiguousarray(val_idxs, dtype=np.int32)
    out = _buf(3 * _cb() * (num_columns // 2))
    lib().oracle_smvp_signed(col_ptr.ctypes.data, val_idxs.ctypes.data, points_xy, len(points_xy) // (2 * _cb()), num_columns, out)
    return out.raw


def bucket_reduction(kind, buckets_xyz, num_threads=1):
    code = {"serial": 0, "running_sum": 1, "parallel": 2}[kind]
    out = _buf(3 * _cb())
    lib().oracle_bucket_reduction(code, buckets_xyz, len(buckets_xyz) // (3 * _cb()), num_threads, out)
    return out.raw


def parallel_bucket_reduction_1(buckets_xyz, num_threads):
    g, m = _buf(3 * _cb() * num_threads), _buf(3 * _cb() * num_threads)
    lib().oracle_parallel_bucket_reduction_1(buckets_xyz, len(buckets_xyz) // (3 * _cb()), num_threads, g, m)
    return g.raw, m.raw


def parallel_bucket_reduction_2(g, m, num_buckets, num_threads):
    out = _buf(3 * _cb() * num_threads)
    lib().oracle_parallel_bucket_reduction_2(g, m, num_buckets, num_threads, out)
    return out.raw


def horner(window_sums_xyz, word_size=16):
    out = _buf(3 * _cb())
    lib().oracle_horner(window_sums_xyz, len(window_sums_xyz) // (3 * _cb()), word_size, out)
    return out.raw


def msm_cuzk_model(points_xy, scalars, word_size=16):
    out = _buf(3 * _cb())
    rc = lib().oracle_msm_cuzk_model(points_xy, scalars, len(scalars) // 32, word_size, out)
    if rc != 0:
        raise ValueError("final carry is 1")
    return out.raw


def sample_scalars(seed, n, first=0):
    out = _buf(32 * n)
    lib().oracle_sample_scalars(seed, first, n, out)
    return out.raw


def sample_points(seed, n, first=0):
    out = _buf(2 * _cb() * n)
    lib().oracle_sample_points(seed, first, n, out)
    return out.raw


def constants():
    p, r, r2, one = _buf(_cb()), _buf(32), _buf(_cb()), _buf(_cb())
    n0 = C.c_uint64()
    lib().oracle_constants(p, r, r2, one, C.byref(n0))
    le = lambda b: int.from_bytes(b.raw, "little")
    return {"p": le(p), "r": le(r), "R2_mod_p": le(r2), "R_mod_p": le(one), "n0inv64": n0.value}
