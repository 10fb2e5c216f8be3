"""ctypes binding of libmsm_hip.so and the Python mirror of the reference's Rust API for the MSM path.

Reference surface mirrored here (paths relative to /root/reference):
    run_webgpu_msm(g, v) -> C::Curve          src/lib.rs:76-82
    compute_msm(points, scalars) -> C::Curve  src/cuzk/msm.rs:75-417
    points_to_bytes / scalars_to_bytes        src/lib.rs:50-65
    sample_points / sample_scalars            src/lib.rs:20-42   (seeded here; the reference uses thread_rng)
Error behaviour: the reference panics (gpu.rs:22,51; lib.rs:58; msm.rs:399; utils.rs:20); here every failure raises
MsmHipError carrying the C-ABI error code.
"""
import ctypes as C
import os

import numpy as np
import torch

from . import build as _build

P = 21888242871839275222246405745257275088696311157297823662689037894645226208583  # src/cuzk/msm.rs:39
R_BN254 = 21888242871839275222246405745257275088548364400416034343698204186575808495617  # BN254's scalar field = Grumpkin's base field
# curve of a context (include/msm_hip.h: MSM_HIP_CURVE_*): id and base-field modulus
PALLAS_P = 0x40000000000000000000000000000000224698fc094cf91b992d30ed00000001  # Pallas' base field = Vesta's scalar field
VESTA_P = 0x40000000000000000000000000000000224698fc0994a8dd8c46eb2100000001   # Vesta's base field = Pallas' scalar field
BLS12_381_P = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
CURVES = {"bn254": (0, P), "grumpkin": (1, R_BN254), "pallas": (2, PALLAS_P), "vesta": (3, VESTA_P), "bls12_381": (4, BLS12_381_P),
          "bn254_g2": (5, P), "bls12_381_g2": (6, BLS12_381_P)}  # G2: coordinates in Fq2 = Fq[u] / (u^2 + 1), an element on the wire is c0 || c1


def coord_bytes(curve):
    """Bytes of a coordinate on a curve's wire: 32; 48 for BLS12-381; 64 / 96 for BN254 / BLS12-381 G2 (an Fq2 element) -- points 2 x, Jacobian
    records 3 x that."""
    return {"bls12_381": 48, "bn254_g2": 64, "bls12_381_g2": 96}.get(curve, 32)
NUM_WINDOWS = 16
WINDOW_BITS = 16
BUCKETS_PER_WINDOW = 1 << 15

_lib = None


class MsmHipError(RuntimeError):
    def __init__(self, code, where):
        self.code = code
        msg = lib().msm_hip_strerror(code).decode() if _lib is not None else "error"
        super().__init__("%s failed: %s (%d)" % (where, msg, code))


def lib():
    """Load libmsm_hip.so (in-tree).  Raises if it has not been built -- there is no fallback path."""
    global _lib
    if _lib is None:
        so = _build.SO
        if not os.path.exists(so):
            raise ImportError("libmsm_hip.so is not built (%s); run `python -c 'import __graft_entry__ as g; g.build()'`" % so)
        L = C.CDLL(so)
        vp, u8p, sz, i = C.c_void_p, C.c_char_p, C.c_size_t, C.c_int
        L.msm_hip_strerror.restype = C.c_char_p
        L.msm_hip_strerror.argtypes = [i]
        L.msm_hip_abi_version.restype = i
        L.msm_hip_ctx_create.argtypes = [C.POINTER(vp), i]
        L.msm_hip_ctx_create_curve.argtypes = [C.POINTER(vp), i, i]
        L.msm_hip_ctx_curve.argtypes = [vp]
        L.msm_hip_combine_windows_curve.argtypes = [i, u8p, i, u8p]
        L.msm_hip_g1_to_affine_curve.argtypes = [i, u8p, u8p]
        L.msm_hip_ctx_destroy.argtypes = [vp]
        L.msm_hip_ctx_destroy.restype = None
        L.msm_hip_set_bases.argtypes = [vp, u8p, sz, C.c_uint32]
        L.msm_hip_set_bases_device.argtypes = [vp, vp, sz, C.c_uint32]
        L.msm_hip_run.argtypes = [vp, u8p, sz, u8p]
        L.msm_hip_run_device.argtypes = [vp, vp, sz, u8p]
        L.msm_hip_run_batch.argtypes = [vp, u8p, sz, sz, u8p]
        L.msm_hip_launch_windows_batch_device.argtypes = [vp, vp, sz, i, i, i, i, vp]
        L.msm_hip_finish_batch.argtypes = [vp, i, u8p]
        L.msm_hip_launch_half_windows_batch_device.argtypes = [vp, vp, sz, i, i, i, i, vp]
        L.msm_hip_combine_windows_batch_curve.argtypes = [i, vp, i, i, u8p]
        L.msm_hip_launch_vwindows_batch_device.argtypes = [vp, vp, sz, i, i, i, i, vp]
        L.msm_hip_combine_vwindows_batch_curve.argtypes = [i, vp, i, i, u8p]
        L.msm_hip_mgpu_set_wide_bits.argtypes = [vp, i]
        L.msm_hip_run_batch_device.argtypes = [vp, vp, sz, sz, u8p]
        L.msm_hip_launch_device.argtypes = [vp, vp, sz, i]
        L.msm_hip_launch.argtypes = [vp, u8p, sz, i]
        L.msm_hip_wait_stream.argtypes = [vp, vp]
        L.msm_hip_finish.argtypes = [vp, i, u8p]
        L.msm_hip_run_windows_device.argtypes = [vp, vp, sz, i, i, vp]
        L.msm_hip_launch_windows_device.argtypes = [vp, vp, sz, i, i, i, vp]
        L.msm_hip_slot_wait_stream.argtypes = [vp, i, vp]
        L.msm_hip_slot_sync.argtypes = [vp, i]
        L.msm_hip_combine_windows_bn254.argtypes = [u8p, i, u8p]
        L.msm_hip_msm_bn254_g1.argtypes = [u8p, u8p, sz, u8p]
        L.msm_hip_msm_curve.argtypes = [i, u8p, u8p, sz, u8p]
        L.msm_hip_test_oneshot_parts.argtypes = [i, sz]
        L.msm_hip_oneshot_release.argtypes = []
        L.msm_hip_oneshot_release.restype = None
        L.msm_hip_sample_scalars_device.argtypes = [vp, C.c_uint64, sz, vp]
        L.msm_hip_sample_points_device.argtypes = [vp, C.c_uint64, sz, vp]
        L.msm_hip_last_stage_ms.argtypes = [vp, C.POINTER(C.c_float), i]
        L.msm_hip_stream.argtypes = [vp]
        L.msm_hip_stream.restype = vp
        L.msm_hip_set_debug.argtypes = [vp, i]
        L.msm_hip_set_fine_hist_min_n.argtypes = [vp, sz]
        L.msm_hip_set_scalar_format.argtypes = [vp, C.c_uint32]
        L.msm_hip_set_stage_timing.argtypes = [vp, i]
        L.msm_hip_set_window_bits.argtypes = [vp, i]
        L.msm_hip_set_wide_bits.argtypes = [vp, i]
        L.msm_hip_wide_bits.argtypes = [vp]
        L.msm_hip_wide_config.argtypes = [i, i, C.c_size_t] + [C.POINTER(C.c_int)] * 4
        L.msm_hip_window_config.argtypes = [i, C.POINTER(i), C.POINTER(i)]
        L.msm_hip_last_window_bits.argtypes = [vp]
        L.msm_hip_endomorphism_window_count.argtypes = [i]
        L.msm_hip_uses_endomorphism.argtypes = [vp]
        L.msm_hip_batch_group_size.argtypes = [vp, sz]
        L.msm_hip_read_digits.argtypes = [vp, vp, sz]
        L.msm_hip_read_col_ptr.argtypes = [vp, vp, sz]
        L.msm_hip_read_val_idxs.argtypes = [vp, vp, sz]
        L.msm_hip_read_buckets.argtypes = [vp, vp, sz]
        L.msm_hip_read_window_sums.argtypes = [vp, vp, sz]
        L.msm_hip_test_fq_op.argtypes = [vp, i, u8p, u8p, u8p, sz]
        L.msm_hip_test_g1_op.argtypes = [vp, i, u8p, u8p, u8p, sz]
        L.msm_hip_test_g1_mul_u32.argtypes = [vp, u8p, vp, u8p, sz]
        L.msm_hip_last_hip_error.argtypes = [vp]
        L.msm_hip_mgpu_create.argtypes = [C.POINTER(vp), C.POINTER(i), i, C.c_uint32]
        L.msm_hip_mgpu_create_curve.argtypes = [C.POINTER(vp), C.POINTER(i), i, C.c_uint32, i]
        L.msm_hip_mgpu_destroy.argtypes = [vp]
        L.msm_hip_mgpu_destroy.restype = None
        L.msm_hip_mgpu_device_count.argtypes = [vp]
        L.msm_hip_mgpu_uses_rccl.argtypes = [vp]
        L.msm_hip_mgpu_set_bases.argtypes = [vp, u8p, sz, C.c_uint32]
        L.msm_hip_mgpu_run.argtypes = [vp, u8p, sz, u8p]
        L.msm_hip_mgpu_run_batch.argtypes = [vp, u8p, sz, sz, u8p]
        L.msm_hip_mgpu_launch_batch.argtypes = [vp, u8p, sz, i, i]
        L.msm_hip_mgpu_launch_batch_device.argtypes = [vp, C.POINTER(vp), sz, i, i]
        L.msm_hip_mgpu_finish_batch.argtypes = [vp, i, u8p]
        L.msm_hip_mgpu_group_size.argtypes = [vp]
        L.msm_hip_mgpu_inject_fault.argtypes = [vp, i, i]
        L.msm_hip_window_range.argtypes = [i, i, i, C.POINTER(i), C.POINTER(i)]
        _lib = L
    return _lib


def _check(code, where):
    if code != 0:
        raise MsmHipError(code, where)


# ------------------------------------------------------------------------------------------------ wire format
def points_to_bytes(points):
    """[(x, y), ...] canonical integers -> n x 64 B, x || y little-endian (src/lib.rs:55-65).
    The point at infinity (None) is not representable: the reference panics at lib.rs:58, this raises."""
    out = bytearray()
    for pt in points:
        if pt is None:
            raise ValueError("point at infinity has no coordinates (src/lib.rs:58)")
        x, y = pt
        if not (0 <= x < P and 0 <= y < P):
            raise ValueError("coordinate out of range")
        out += int(x).to_bytes(32, "little") + int(y).to_bytes(32, "little")
    return bytes(out)


def scalars_to_bytes(scalars):
    """[s, ...] integers in [0, r) -> n x 32 B little-endian (src/lib.rs:50-52)."""
    return b"".join(int(s).to_bytes(32, "little") for s in scalars)


class G1:
    """Result of an MSM: a Jacobian point (x, y, z), canonical integers, z = 0 <=> identity (≙ C::Curve)."""

    __slots__ = ("xyz", "p")

    def __init__(self, xyz, p=P):
        self.xyz = bytes(xyz)
        self.p = p  # base-field modulus of the point's curve
        assert len(self.xyz) in (96, 144, 192, 288)  # 3 coordinates of 32 bytes (48: BLS12-381; 64 / 96: BN254 / BLS12-381 G2, coordinates in Fq2)

    @property
    def quadratic(self):
        """The coordinates are Fq2 elements (c0, c1) (a G2 point: 192-byte record; 288 bytes on BLS12-381)."""
        return len(self.xyz) in (192, 288)

    def coords(self):
        b, cb = self.xyz, len(self.xyz) // 3
        if self.quadratic:
            h = cb // 2
            return tuple((int.from_bytes(b[k:k + h], "little"), int.from_bytes(b[k + h:k + cb], "little")) for k in (0, cb, 2 * cb))
        return tuple(int.from_bytes(b[k:k + cb], "little") for k in (0, cb, 2 * cb))

    def is_identity(self):
        return self.coords()[2] in (0, (0, 0))

    def to_affine(self):
        """(x, y) canonical integers (pairs (c0, c1) for a G2 point), or None for the identity (≙ Curve::to_affine, tests/cuzk.rs:88-94)."""
        x, y, z = self.coords()
        p = self.p
        if self.quadratic:
            if z == (0, 0):
                return None
            mul = lambda a, b: ((a[0] * b[0] - a[1] * b[1]) % p, (a[0] * b[1] + a[1] * b[0]) % p)
            n = pow(z[0] * z[0] + z[1] * z[1], -1, p)
            zi = (z[0] * n % p, -z[1] * n % p)
            zi2 = mul(zi, zi)
            return (mul(x, zi2), mul(y, mul(zi2, zi)))
        if z == 0:
            return None
        zi = pow(z, -1, p)
        return (x * zi * zi % p, y * zi * zi * zi % p)

    def to_affine_bytes(self):
        """The canonical affine encoding x || y used for bit-exact comparison (64 bytes; 96 for BLS12-381, 128 for G2); zero bytes for the identity."""
        a, cb = self.to_affine(), len(self.xyz) // 3
        if a is None:
            return bytes(2 * cb)
        if self.quadratic:
            return b"".join(c.to_bytes(cb // 2, "little") for c in (a[0][0], a[0][1], a[1][0], a[1][1]))
        return a[0].to_bytes(cb, "little") + a[1].to_bytes(cb, "little")

    def __eq__(self, other):  # projective equality, as G1's PartialEq (src/lib.rs:166)
        return isinstance(other, G1) and self.to_affine() == other.to_affine()

    def __hash__(self):
        return hash(self.to_affine())

    def __repr__(self):
        return "G1(%s)" % (self.to_affine(),)


# ------------------------------------------------------------------------------------------------ context
def _as_device_u8(t, row, what):
    if not (isinstance(t, torch.Tensor) and t.is_cuda):
        raise TypeError("%s must be a CUDA(HIP) uint8 tensor" % what)
    if t.dtype != torch.uint8 or not t.is_contiguous() or t.numel() % row:
        raise ValueError("%s must be contiguous uint8 with a multiple of %d bytes" % (what, row))
    return t, t.numel() // row


class MsmContext:
    """Persistent engine on one GPU: stream, pooled buffers, resident bases (include/msm_hip.h)."""

    def __init__(self, device=0, curve="bn254"):
        self._h = C.c_void_p()
        self.curve = curve
        self.curve_id, self.modulus = CURVES[curve]
        self.cb = coord_bytes(curve)  # bytes per coordinate; a point is pb = 2 cb, a Jacobian record jb = 3 cb
        self.pb, self.jb = 2 * self.cb, 3 * self.cb
        _check(lib().msm_hip_ctx_create_curve(C.byref(self._h), int(device), self.curve_id), "msm_hip_ctx_create_curve")
        self.device = int(device)
        self.n_bases = 0
        self.wide_bits_choice = 0
        self._keepalive = {}  # slot -> tensors the slot's launch still reads / writes; released when the slot is collected

    def _order_after_torch(self, *tensors):
        """The engine's streams are not ordered with torch's: make its main stream wait (on the device) for everything
        enqueued so far on the torch stream that produced the inputs (include/msm_hip.h, ordering contract)."""
        for t in tensors:
            if isinstance(t, torch.Tensor) and t.is_cuda:
                _check(lib().msm_hip_wait_stream(self._h, torch.cuda.current_stream(t.device).cuda_stream), "msm_hip_wait_stream")
                return

    def close(self):
        if self._h:
            lib().msm_hip_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- bases
    def set_bases(self, points, check_on_curve=False, mont256=False, precompute=False, endomorphism=False):
        """points: bytes (host, n x 64 B wire format) or a CUDA uint8 tensor holding the same bytes.
        mont256: the coordinates are x * 2^256 mod p (4 x 64-bit Montgomery limbs, R = 2^256) instead of canonical integers.
        precompute: fixed-base tables 2^(16 w) P_i (16 x the memory): whole MSMs then use one bucket set for all windows.  "wide":
        MSM_HIP_BASES_PRECOMPUTE_WIDE -- tables 2^(C w) P_i for digits of C = 17 (more than 2^20 bases: 20; up to 2^16: 16) bits: 15 (13) bucket additions per
        point instead of 16, one bucket set (large MSMs; set_wide_bits overrides C).
        endomorphism: True: also store phi(P_i) (2 x the memory): whole MSMs split every scalar into two 127-bit halves and need
        half the windows.  False (this wrapper's default: the stage-level parity tests read the reference's 16-window shape):
        MSM_HIP_BASES_PLAIN.  None: the C ABI's own default (flags = 0) -- the fastest mode the curve has, which is what the
        reference-shaped calls (compute_msm / run_webgpu_msm below, msm_hip_msm_bn254_g1) use."""
        flags = (1 if check_on_curve else 0) | (2 if mont256 else 0) | (32 if precompute == "wide" else 4 if precompute else 0) | (8 if endomorphism else 0)
        if endomorphism is False and not precompute:
            flags |= 16
        if isinstance(points, torch.Tensor) and points.is_cuda:
            t, n = _as_device_u8(points, self.pb, "points")
            self._order_after_torch(t)
            _check(lib().msm_hip_set_bases_device(self._h, t.data_ptr(), n, flags), "msm_hip_set_bases_device")
        else:
            b = bytes(points)
            if len(b) % self.pb:
                raise ValueError("points must be n x 64 bytes")
            n = len(b) // self.pb
            _check(lib().msm_hip_set_bases(self._h, b, n, flags), "msm_hip_set_bases")
        self.n_bases = n
        return n

    # -- whole MSM
    def msm(self, scalars):
        """sum_i scalars[i] * bases[i] -> G1.  scalars: bytes (n x 32 B) or CUDA uint8 tensor."""
        out = C.create_string_buffer(self.jb)
        if isinstance(scalars, torch.Tensor) and scalars.is_cuda:
            t, n = _as_device_u8(scalars, 32, "scalars")
            self._order_after_torch(t)
            _check(lib().msm_hip_run_device(self._h, t.data_ptr(), n, out), "msm_hip_run_device")
        else:
            b = bytes(scalars)
            if len(b) % 32:
                raise ValueError("scalars must be n x 32 bytes")
            _check(lib().msm_hip_run(self._h, b, len(b) // 32, out), "msm_hip_run")
        return G1(out.raw, self.modulus)

    def msm_batch(self, scalars_dev, n):
        """`batch` MSMs over the resident bases: scalars_dev is a CUDA uint8 tensor of batch x n x 32 bytes (or host bytes
        of the same layout) -> [G1, ...]."""
        if isinstance(scalars_dev, (bytes, bytearray)):
            b = bytes(scalars_dev)
            if n <= 0 or len(b) % (32 * n):
                raise ValueError("scalars must hold a whole number of n-element vectors")
            batch = len(b) // (32 * n)
            out = C.create_string_buffer(self.jb * batch)
            _check(lib().msm_hip_run_batch(self._h, b, n, batch, out), "msm_hip_run_batch")
            return [G1(out.raw[self.jb * k:self.jb * (k + 1)], self.modulus) for k in range(batch)]
        t, rows = _as_device_u8(scalars_dev, 32, "scalars")
        if n <= 0 or rows % n:
            raise ValueError("scalars must hold a whole number of n-element vectors")
        batch = rows // n
        out = C.create_string_buffer(self.jb * batch)
        self._order_after_torch(t)
        _check(lib().msm_hip_run_batch_device(self._h, t.data_ptr(), n, batch, out), "msm_hip_run_batch_device")
        return [G1(out.raw[self.jb * k:self.jb * (k + 1)], self.modulus) for k in range(batch)]

    def launch(self, scalars_dev, slot=0):
        """Enqueue the device work of one MSM into a result slot (0..3) and return at once."""
        t, n = _as_device_u8(scalars_dev, 32, "scalars")
        self._order_after_torch(t)
        _check(lib().msm_hip_launch_device(self._h, t.data_ptr(), n, slot), "msm_hip_launch_device")
        self._keepalive[slot] = t

    def launch_host(self, scalars_host, slot=0):
        """`launch` with the scalars in host memory (bytes, n x 32 B): copied on the engine's copy stream into the slot's own
        staging buffer, so the copy of the next MSM overlaps the device work of the current one when slots alternate."""
        b = bytes(scalars_host)
        if len(b) % 32:
            raise ValueError("scalars must be n x 32 bytes")
        _check(lib().msm_hip_launch(self._h, b, len(b) // 32, slot), "msm_hip_launch")

    def finish(self, slot=0):
        """Wait for the slot's device work, run the host window combine, return G1."""
        out = C.create_string_buffer(self.jb)
        try:
            _check(lib().msm_hip_finish(self._h, slot, out), "msm_hip_finish")
        finally:
            self._keepalive.pop(slot, None)
        return G1(out.raw, self.modulus)

    # -- window shard (multi-GPU)
    def msm_windows(self, scalars_dev, w_begin, w_end, out_dev=None):
        """Window sums S_w, w in [w_begin, w_end), as a CUDA uint8 tensor [(w_end - w_begin), 96]."""
        t, n = _as_device_u8(scalars_dev, 32, "scalars")
        if out_dev is None:
            out_dev = torch.empty((w_end - w_begin, self.jb), dtype=torch.uint8, device=t.device)
        self._order_after_torch(t)
        _check(lib().msm_hip_run_windows_device(self._h, t.data_ptr(), n, w_begin, w_end, out_dev.data_ptr()),
               "msm_hip_run_windows_device")
        return out_dev

    def launch_windows(self, scalars_dev, w_begin, w_end, slot, out_dev):
        """Asynchronous msm_windows into a result slot (0..3); `out_dev` (CUDA uint8 [w_end - w_begin, 96]) receives the sums."""
        t, n = _as_device_u8(scalars_dev, 32, "scalars")
        self._order_after_torch(t)
        _check(lib().msm_hip_launch_windows_device(self._h, t.data_ptr(), n, w_begin, w_end, slot, out_dev.data_ptr()),
               "msm_hip_launch_windows_device")
        self._keepalive[slot] = (t, out_dev)

    def launch_windows_batch(self, scalars_dev, n, w_begin, w_end, slot, out_dev, inputs_complete=False):
        """Several MSMs per launch: scalars_dev holds nvec contiguous vectors of n scalars (CUDA uint8 [nvec * n, 32]);
        `out_dev` (CUDA uint8 [nvec * (w_end - w_begin), 96], vector-major) receives the window sums.  nvec * windows <= 64; out_dev None keeps the sums in the slot (whole MSMs: finish_batch)."""
        t, rows = _as_device_u8(scalars_dev, 32, "scalars")
        if n <= 0 or rows % n:
            raise ValueError("scalars must hold a whole number of n-element vectors")
        if not inputs_complete:  # True: the caller vouches that the scalars were complete before this call (no stream ordering needed)
            self._order_after_torch(t)
        _check(lib().msm_hip_launch_windows_batch_device(self._h, t.data_ptr(), n, rows // n, w_begin, w_end, slot,
                                                               out_dev.data_ptr() if out_dev is not None else None),
               "msm_hip_launch_windows_batch_device")
        self._keepalive[slot] = (t, out_dev)

    def launch_half_windows_batch(self, scalars_dev, n, hw_begin, hw_end, slot, out_dev, inputs_complete=False):
        """launch_windows_batch over the 8 HALF-length windows of a context whose bases carry their endomorphism images
        (msm_hip_launch_half_windows_batch_device): `out_dev` receives nvec * (hw_end - hw_begin) sums, vector-major."""
        t, rows = _as_device_u8(scalars_dev, 32, "scalars")
        if n <= 0 or rows % n:
            raise ValueError("scalars must hold a whole number of n-element vectors")
        if not inputs_complete:
            self._order_after_torch(t)
        _check(lib().msm_hip_launch_half_windows_batch_device(self._h, t.data_ptr(), n, rows // n, hw_begin, hw_end, slot,
                                                                    out_dev.data_ptr() if out_dev is not None else None),
               "msm_hip_launch_half_windows_batch_device")
        self._keepalive[slot] = (t, out_dev)

    def launch_vwindows_batch(self, scalars_dev, n, v_begin, v_end, slot, out_dev, inputs_complete=False):
        """launch_windows_batch over the VIRTUAL windows of a context whose bases are wide fixed-base tables (set_bases(precompute="wide");
        msm_hip_launch_vwindows_batch_device): `out_dev` (CUDA uint8 [nvec * (v_end - v_begin) * 2, jb], vector-major) receives, for every
        virtual window, its weighted sum and its plain total."""
        t, rows = _as_device_u8(scalars_dev, 32, "scalars")
        if n <= 0 or rows % n:
            raise ValueError("scalars must hold a whole number of n-element vectors")
        if not inputs_complete:
            self._order_after_torch(t)
        _check(lib().msm_hip_launch_vwindows_batch_device(self._h, t.data_ptr(), n, rows // n, v_begin, v_end, slot,
                                                          out_dev.data_ptr() if out_dev is not None else None),
               "msm_hip_launch_vwindows_batch_device")
        self._keepalive[slot] = (t, out_dev)

    def launch_batch(self, scalars_dev, n, slot=0):
        """Enqueue up to 4 WHOLE MSMs (contiguous scalar vectors, CUDA uint8 [nvec * n, 32]) as one launch; finish_batch collects."""
        self.launch_windows_batch(scalars_dev, n, 0, NUM_WINDOWS, slot, None)
        return scalars_dev.numel() // (32 * n)

    def finish_batch(self, slot, nvec):
        """Wait for a launch_batch slot, run the host window combines, return the list of G1 results."""
        out = C.create_string_buffer(self.jb * nvec)
        try:
            _check(lib().msm_hip_finish_batch(self._h, slot, out), "msm_hip_finish_batch")
        finally:
            self._keepalive.pop(slot, None)
        return [G1(out.raw[self.jb * k:self.jb * (k + 1)], self.modulus) for k in range(nvec)]

    def slot_wait_stream(self, slot, stream=None):
        """Make a torch CUDA stream (default: the current one) wait, on the device, for the slot's results."""
        if stream is None:
            stream = torch.cuda.current_stream()
        _check(lib().msm_hip_slot_wait_stream(self._h, slot, stream.cuda_stream), "msm_hip_slot_wait_stream")

    def slot_sync(self, slot):
        """Block until the slot is complete; raises on a device-side input error."""
        try:
            _check(lib().msm_hip_slot_sync(self._h, slot), "msm_hip_slot_sync")
        finally:
            self._keepalive.pop(slot, None)

    @staticmethod
    def combine_windows(window_sums, curve="bn254"):
        """Host Horner over all window sums (bytes or uint8 tensor, num_windows x 96 B) -> G1."""
        if isinstance(window_sums, torch.Tensor):
            window_sums = window_sums.cpu().contiguous().numpy().tobytes()
        b = bytes(window_sums)
        jb = 3 * coord_bytes(curve)
        out = C.create_string_buffer(jb)
        cid, p = CURVES[curve]
        _check(lib().msm_hip_combine_windows_curve(cid, b, len(b) // jb, out), "msm_hip_combine_windows_curve")
        return G1(out.raw, p)

    @staticmethod
    def combine_windows_batch(window_sums, num_windows, curve="bn254"):
        """Host Horner for several MSMs at once: window_sums holds nvec x num_windows x 96 B (bytes, numpy uint8 array or CPU tensor);
        the chains run side by side on the library's host pool.  -> [G1, ...]"""
        if isinstance(window_sums, torch.Tensor):
            window_sums = window_sums.contiguous().numpy()
        a = np.ascontiguousarray(np.frombuffer(window_sums, dtype=np.uint8) if isinstance(window_sums, (bytes, bytearray)) else window_sums, dtype=np.uint8)
        jb = 3 * coord_bytes(curve)
        nvec = a.size // (jb * num_windows)
        if nvec * jb * num_windows != a.size:
            raise ValueError("window sums must be nvec x num_windows x %d bytes" % jb)
        out = C.create_string_buffer(max(jb * nvec, 1))
        cid, p = CURVES[curve]
        _check(lib().msm_hip_combine_windows_batch_curve(cid, a.ctypes.data, num_windows, nvec, out), "msm_hip_combine_windows_batch_curve")
        return [G1(out.raw[jb * k:jb * (k + 1)], p) for k in range(nvec)]

    @staticmethod
    def combine_vwindows_batch(pairs, num_vwindows, curve="bn254"):
        """The finish of window-sharded launches over wide tables: pairs holds nvec x num_vwindows x 2 records (bytes, numpy uint8 array or
        CPU tensor) -- every virtual window's weighted sum and plain total, in virtual-window order -> [G1, ...]"""
        if isinstance(pairs, torch.Tensor):
            pairs = pairs.contiguous().numpy()
        a = np.ascontiguousarray(np.frombuffer(pairs, dtype=np.uint8) if isinstance(pairs, (bytes, bytearray)) else pairs, dtype=np.uint8)
        jb = 3 * coord_bytes(curve)
        nvec = a.size // (2 * jb * num_vwindows)
        if nvec * 2 * jb * num_vwindows != a.size:
            raise ValueError("pairs must be nvec x num_vwindows x 2 x %d bytes" % jb)
        out = C.create_string_buffer(max(jb * nvec, 1))
        cid, p = CURVES[curve]
        _check(lib().msm_hip_combine_vwindows_batch_curve(cid, a.ctypes.data, num_vwindows, nvec, out), "msm_hip_combine_vwindows_batch_curve")
        return [G1(out.raw[jb * k:jb * (k + 1)], p) for k in range(nvec)]

    def virtual_windows(self):
        """virtual windows (of 2^15 bucket slots) of the resident wide tables: 2^(digit bits - 16); 0 without such tables"""
        wb = self.wide_bits()
        return (1 << (wb - 16)) if wb else 0

    # -- synthetic inputs in HBM
    def sample_scalars(self, n, seed):
        t = torch.empty((n, 32), dtype=torch.uint8, device="cuda:%d" % self.device)
        _check(lib().msm_hip_sample_scalars_device(self._h, seed, n, t.data_ptr()), "msm_hip_sample_scalars_device")
        return t

    def sample_points(self, n, seed):
        t = torch.empty((n, self.pb), dtype=torch.uint8, device="cuda:%d" % self.device)
        _check(lib().msm_hip_sample_points_device(self._h, seed, n, t.data_ptr()), "msm_hip_sample_points_device")
        return t

    # -- measurement
    def set_stage_timing(self, level):
        """0: no stage events; 1: only around the SMVP accumulate kernel; 2: every stage boundary (default)."""
        _check(lib().msm_hip_set_stage_timing(self._h, int(level)), "msm_hip_set_stage_timing")

    def stage_ms(self):
        buf = (C.c_float * 10)()
        k = lib().msm_hip_last_stage_ms(self._h, buf, 10)
        names = ["recode_count", "coarse_scan", "coarse_scatter", "fine_sort", "smvp", "smvp_stitch", "bucket_reduce",
                 "device_total", "host_finalise"]
        return {names[j]: float(buf[j]) for j in range(k)}

    # -- window size
    def set_wide_bits(self, bits):
        """digit width (17 .. 20; 0: by the number of bases) of the wide fixed-base tables the next set_bases(precompute="wide") builds"""
        _check(lib().msm_hip_set_wide_bits(self._h, int(bits)), "msm_hip_set_wide_bits")
        self.wide_bits_choice = int(bits)

    def wide_bits(self):
        """digit width of the resident wide tables (0: the bases are not held that way)"""
        return lib().msm_hip_wide_bits(self._h)

    def set_window_bits(self, bits):
        """0: whole-MSM launches pick the signed-digit window from n (12 / 14 / 16 bits); 12, 14 or 16 fixes it."""
        _check(lib().msm_hip_set_window_bits(self._h, int(bits)), "msm_hip_set_window_bits")

    def batch_group_size(self, n):
        """Whole MSMs of n points per launch (launch_batch) for best throughput."""
        return lib().msm_hip_batch_group_size(self._h, n)

    def last_window_bits(self):
        return lib().msm_hip_last_window_bits(self._h)

    @staticmethod
    def window_config(bits):
        """(number of windows, buckets per window) of a window size."""
        nw, nb = C.c_int(), C.c_int()
        _check(lib().msm_hip_window_config(int(bits), C.byref(nw), C.byref(nb)), "msm_hip_window_config")
        return nw.value, nb.value

    @staticmethod
    def endomorphism_window_count(bits):
        """Windows of one 127-bit half at a window size (8 at 16 bits)."""
        v = lib().msm_hip_endomorphism_window_count(int(bits))
        if v < 0:
            raise MsmHipError(v, "msm_hip_endomorphism_window_count")
        return v

    def uses_endomorphism(self):
        return lib().msm_hip_uses_endomorphism(self._h) == 1

    # -- stage read-back (parity tests)
    def set_scalar_format(self, mont256):
        """False: canonical little-endian scalars (default); True: s * 2^256 mod r words (R = 2^256 Montgomery limbs)."""
        _check(lib().msm_hip_set_scalar_format(self._h, 1 if mont256 else 0), "msm_hip_set_scalar_format")

    def set_fine_hist_min_n(self, n):
        _check(lib().msm_hip_set_fine_hist_min_n(self._h, n), "msm_hip_set_fine_hist_min_n")

    def set_debug(self, keep_digit_planes=True):
        _check(lib().msm_hip_set_debug(self._h, 1 if keep_digit_planes else 0), "msm_hip_set_debug")

    def read_digits(self, n, w_count=NUM_WINDOWS):
        a = np.empty((w_count, n), dtype=np.uint16)
        _check(lib().msm_hip_read_digits(self._h, a.ctypes.data, a.size), "msm_hip_read_digits")
        return a

    def read_col_ptr(self, w_count=NUM_WINDOWS, buckets=BUCKETS_PER_WINDOW):
        a = np.empty((w_count, buckets + 1), dtype=np.uint32)
        _check(lib().msm_hip_read_col_ptr(self._h, a.ctypes.data, a.size), "msm_hip_read_col_ptr")
        return a

    def read_val_idxs(self, n, w_count=NUM_WINDOWS):
        a = np.empty((w_count, n), dtype=np.uint32)
        _check(lib().msm_hip_read_val_idxs(self._h, a.ctypes.data, a.size), "msm_hip_read_val_idxs")
        return a

    def read_buckets(self, w_count=NUM_WINDOWS, buckets=BUCKETS_PER_WINDOW):
        a = np.empty((w_count, buckets, self.jb), dtype=np.uint8)
        _check(lib().msm_hip_read_buckets(self._h, a.ctypes.data, a.size), "msm_hip_read_buckets")
        return a

    def read_window_sums(self, w_count=NUM_WINDOWS):
        a = np.empty((w_count, self.jb), dtype=np.uint8)
        _check(lib().msm_hip_read_window_sums(self._h, a.ctypes.data, a.size), "msm_hip_read_window_sums")
        return a

    # -- single-op hooks (≙ tests/field.rs, tests/point.rs)
    def fq_op(self, op, a, b=None):
        code = {"add": 0, "sub": 1, "mul": 2, "sqr": 3, "neg": 4, "mul_asm": 5, "sqr_asm": 6, "mul2_asm": 7, "mul_asm_lazy": 8,
                "sqr_asm_lazy": 9}[op]
        n = len(a) // self.cb
        out = C.create_string_buffer(max(self.cb * n, 1))
        _check(lib().msm_hip_test_fq_op(self._h, code, a, b, out, n), "msm_hip_test_fq_op")
        return out.raw[:self.cb * n]

    def g1_op(self, op, a, b=None):
        code = {"add": 0, "double": 1, "add_affine": 2, "madd_w_pmp": 3, "madd_w_mm": 4}[op]
        n = len(a) // self.jb
        out = C.create_string_buffer(max(self.jb * n, 1))
        _check(lib().msm_hip_test_g1_op(self._h, code, a, b, out, n), "msm_hip_test_g1_op")
        return out.raw[:self.jb * n]

    def g1_mul_u32(self, a, ks):
        n = len(a) // self.jb
        k = np.ascontiguousarray(ks, dtype=np.uint32)
        out = C.create_string_buffer(max(self.jb * n, 1))
        _check(lib().msm_hip_test_g1_mul_u32(self._h, a, k.ctypes.data, out, n), "msm_hip_test_g1_mul_u32")
        return out.raw[:self.jb * n]


class MultiGpuMsm:
    """Several GPUs driven by ONE process through the C ABI (msm_hip_mgpu_*, include/msm_hip.h): windows of one MSM sharded
    over the devices + gather (RCCL or pinned buffers) + one host window combine; batches dealt out as whole MSMs.
    (bench.py's multi-GPU path is the other shape: one process per GPU over torch.distributed, msm-webgpu_amd/sharding.py.)"""

    GATHER = {"auto": 0, "host": 1, "rccl": 2}

    def __init__(self, device_ids, gather="auto", curve="bn254"):
        ids = (C.c_int * len(device_ids))(*[int(d) for d in device_ids])
        self._h = C.c_void_p()
        self.curve = curve
        self.modulus = CURVES[curve][1]
        self.cb = coord_bytes(curve)
        self.pb, self.jb = 2 * self.cb, 3 * self.cb
        _check(lib().msm_hip_mgpu_create_curve(C.byref(self._h), ids, len(device_ids), self.GATHER[gather], CURVES[curve][0]), "msm_hip_mgpu_create_curve")

    def close(self):
        if self._h:
            lib().msm_hip_mgpu_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def device_count(self):
        return lib().msm_hip_mgpu_device_count(self._h)

    @property
    def uses_rccl(self):
        return lib().msm_hip_mgpu_uses_rccl(self._h) == 1

    def set_bases(self, points, check_on_curve=False, endomorphism=False, precompute=False):
        """Replicated on every device.  endomorphism: True: MSM_HIP_BASES_ENDOMORPHISM -- msm_batch runs whole MSMs over the 2n points, and
        the window-sharded calls shard the 8 half-length windows; False: MSM_HIP_BASES_PLAIN; None: the C ABI's default (flags = 0: whole
        object resolves it to the plain set).  precompute: True -- the 16-bit fixed-base tables for the whole MSMs of msm_batch (the
        window-sharded calls ignore them); "wide" -- the wide tables: msm_batch runs whole MSMs on them and the window-sharded calls share
        their virtual windows (set_wide_bits; default 19-bit digits: 8 virtual windows)."""
        b = bytes(points)
        flags = (1 if check_on_curve else 0) | (8 if endomorphism else 0) | (32 if precompute == "wide" else 4 if precompute else 0)
        if endomorphism is False and not precompute:
            flags |= 16
        _check(lib().msm_hip_mgpu_set_bases(self._h, b, len(b) // self.pb, flags), "msm_hip_mgpu_set_bases")
        return len(b) // self.pb

    def msm(self, scalars):
        b = bytes(scalars)
        out = C.create_string_buffer(self.jb)
        _check(lib().msm_hip_mgpu_run(self._h, b, len(b) // 32, out), "msm_hip_mgpu_run")
        return G1(out.raw, self.modulus)

    def msm_batch(self, scalars, n):
        b = bytes(scalars)
        batch = len(b) // (32 * n)
        out = C.create_string_buffer(max(self.jb * batch, 1))
        _check(lib().msm_hip_mgpu_run_batch(self._h, b, n, batch, out), "msm_hip_mgpu_run_batch")
        return [G1(out.raw[self.jb * k:self.jb * (k + 1)], self.modulus) for k in range(batch)]

    def set_wide_bits(self, bits):
        """digit width (16 .. 20; 0: 19) of the wide tables the next set_bases(precompute="wide") builds on every device"""
        _check(lib().msm_hip_mgpu_set_wide_bits(self._h, int(bits)), "msm_hip_mgpu_set_wide_bits")

    # -- window-sharded launches of several MSMs, asynchronous (the throughput form)
    @property
    def group_size(self):
        """MSMs per launch that fill a device (msm_hip_mgpu_group_size): 16 / windows per device (8 with endomorphism bases)."""
        return lib().msm_hip_mgpu_group_size(self._h)

    def launch_batch(self, scalars, n, slot=0, inputs_complete=False):
        """`scalars`: nvec x n x 32 B of host bytes (every device uploads them), or a list with one CUDA uint8 tensor per device
        holding the same bytes (inputs_complete: they were synchronised before this call; otherwise every tensor's current torch stream is
        waited for here).  Returns nvec; finish_batch(slot, nvec) collects."""
        if isinstance(scalars, (list, tuple)):
            ts = [_as_device_u8(t, 32, "scalars")[0] for t in scalars]
            rows = ts[0].numel() // 32
            if n <= 0 or rows % n or any(t.numel() != ts[0].numel() for t in ts):
                raise ValueError("every device needs the same whole number of n-element vectors")
            ptrs = (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
            if not inputs_complete:  # the header wants the device scalars complete before the call: wait for whatever torch still has queued on
                for t in ts:          # each tensor's device (the engine's streams are not ordered with torch's)
                    torch.cuda.current_stream(t.device).synchronize()
            _check(lib().msm_hip_mgpu_launch_batch_device(self._h, ptrs, n, rows // n, slot), "msm_hip_mgpu_launch_batch_device")
            self._keep = getattr(self, "_keep", {})
            self._keep[slot] = ts
            return rows // n
        b = bytes(scalars)
        if n <= 0 or len(b) % (32 * n):
            raise ValueError("scalars must hold a whole number of n-element vectors")
        self._keep = getattr(self, "_keep", {})
        self._keep[slot] = b  # the library reads the buffer until finish
        _check(lib().msm_hip_mgpu_launch_batch(self._h, b, n, len(b) // (32 * n), slot), "msm_hip_mgpu_launch_batch")
        return len(b) // (32 * n)

    def inject_fault(self, device_index, launches=1):
        """test hook (msm_hip_mgpu_inject_fault): the next `launches` window-sharded launches fail on that device"""
        _check(lib().msm_hip_mgpu_inject_fault(self._h, device_index, launches), "msm_hip_mgpu_inject_fault")

    def finish_batch(self, slot, nvec):
        out = C.create_string_buffer(self.jb * nvec)
        try:
            _check(lib().msm_hip_mgpu_finish_batch(self._h, slot, out), "msm_hip_mgpu_finish_batch")
        finally:
            getattr(self, "_keep", {}).pop(slot, None)
        return [G1(out.raw[self.jb * k:self.jb * (k + 1)], self.modulus) for k in range(nvec)]


def window_range_abi(rank, world, num=NUM_WINDOWS):
    """msm_hip_window_range: the C ABI's partition (must equal sharding.window_range)."""
    b, e = C.c_int(), C.c_int()
    _check(lib().msm_hip_window_range(rank, world, num, C.byref(b), C.byref(e)), "msm_hip_window_range")
    return b.value, e.value


# ------------------------------------------------------------------------------------------------ reference-shaped functions
_default_ctx = {}


def _ctx(device=0):
    if device not in _default_ctx:
        _default_ctx[device] = MsmContext(device)
    return _default_ctx[device]


def compute_msm(points, scalars, device=0):
    """≙ compute_msm (src/cuzk/msm.rs:75): one-shot MSM including base upload; returns G1."""
    ctx = _ctx(device)
    n = ctx.set_bases(points, endomorphism=None)  # the ABI's default: the curve's fastest mode (same group element as the reference's shape)
    ns = (scalars.numel() if isinstance(scalars, torch.Tensor) else len(scalars)) // 32
    if ns != n:
        raise ValueError("points and scalars differ in length (%d vs %d)" % (n, ns))
    return ctx.msm(scalars)


def run_webgpu_msm(g, v, device=0):
    """≙ run_webgpu_msm (src/lib.rs:76-82); the name is the reference's, the device is an MI355X."""
    return compute_msm(g, v, device)


def sample_scalars(n, seed=0, device=0):
    """≙ sample_scalars (src/lib.rs:20-23), seeded; CUDA uint8 tensor [n, 32] in the wire format."""
    return _ctx(device).sample_scalars(n, seed)


def sample_points(n, seed=0, device=0):
    """≙ sample_points (src/lib.rs:36-42), seeded; CUDA uint8 tensor [n, 64] in the wire format."""
    return _ctx(device).sample_points(n, seed)
