"""Which stage of a VARIANT build (MSM_HIP_SO=...) deviates from the CPU stage models: buckets after SMVP + stitch, window sums after the
bucket reduce, the final result -- for a small MSM with a duplicated and a negated-duplicate point (the doubling / cancellation paths).
usage: MSM_HIP_SO=<variant.so> python tools/variant_stage_diff.py [curve] [n] [bits]      (test infrastructure: uses the oracle)"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import msm_webgpu_amd as m

curve = sys.argv[1] if len(sys.argv) > 1 else "bls12_381"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 257
bits = int(sys.argv[3]) if len(sys.argv) > 3 else 16
cpu = importlib.import_module("oracle.cpu_" + curve) if curve != "bn254" else importlib.import_module("oracle.cpu")
cb = cpu.coord_bytes()
PB, JB = 2 * cb, 3 * cb
points, sc = bytearray(cpu.sample_points(34, n)), bytearray(cpu.sample_scalars(35, n))
dup = os.environ.get("DUP", "1") == "1"
if dup and n > 40:
    points[PB * 20:PB * 21] = points[PB * 21:PB * 22]
    sc[32 * 20:32 * 21] = sc[32 * 21:32 * 22]
points, sc = bytes(points), bytes(sc)
ctx = m.MsmContext(0, curve=curve)
ctx.set_bases(points)
ctx.set_debug(True)
ctx.set_window_bits(bits)
res = ctx.msm(sc)
nwin, nb = m.MsmContext.window_config(bits)
buckets, wsums = ctx.read_buckets(nwin, nb), ctx.read_window_sums(nwin)
digits = cpu.decompose_scalars_signed(sc, nwin, bits)
bad_b = bad_w = 0
for w in range(nwin):
    cp, vi = cpu.transpose(digits[w], 2 * nb)
    want = cpu.smvp_signed(cp, vi, points, 2 * nb)
    got = buckets[w].tobytes()
    diff = [k for k in range(nb) if cpu.to_affine64(got[JB * k:JB * k + JB]) != cpu.to_affine64(want[JB * k:JB * k + JB])]
    if diff:
        bad_b += 1
        print("window", w, "buckets differ at slots", diff[:8], "(digits of the duplicated pair:", digits[w][20], digits[w][21], ")")
        if dup and n > 40 and len(diff) == 1:  # what the wrong bucket holds, if it is a small multiple of the duplicated point
            pt = points[PB * 21:PB * 22]
            k = diff[0]
            got_aff = cpu.to_affine64(got[JB * k:JB * k + JB])
            names = {}
            for mult in (1, 2, 3, 4):
                jac = cpu.g1_scalar_mul(pt, int(mult).to_bytes(32, "little"))
                aff = cpu.to_affine64(jac)
                names[aff] = "%d P" % mult
                y = bytearray(aff)
                fb = cb // (2 if curve.endswith("_g2") else 1)
                neg = bytearray(aff[:cb])
                for c0 in range(cb, 2 * cb, fb):
                    v = int.from_bytes(aff[c0:c0 + fb], "little")
                    neg += ((ctx.modulus - v) % ctx.modulus).to_bytes(fb, "little")
                names[bytes(neg)] = "-%d P" % mult
            names[bytes(2 * cb)] = "identity"
            print("   wrong bucket holds:", names.get(got_aff, "none of +-1..4 P / identity"), "| expected:", names.get(cpu.to_affine64(want[JB * k:JB * k + JB]), "?"))
    if cpu.to_affine64(wsums[w].tobytes()) != cpu.to_affine64(cpu.bucket_reduction("running_sum", want)):
        bad_w += 1
        print("window", w, "window sum differs", "(its buckets were", "wrong)" if diff else "right)")
print("result", "ok" if res.to_affine_bytes() == cpu.to_affine64(cpu.cpu_msm(points, sc)) else "WRONG", "| windows with wrong buckets:", bad_b, "| with wrong sums:", bad_w)
