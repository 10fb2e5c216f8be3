#!/usr/bin/env python3
"""Generate msm-webgpu_amd/csrc/fq29_asm.h: the three Montgomery multipliers of fq29.h (fq_mul, fq_mul2, fq_sqr) as ONE inline
assembly block each for gfx950.

Why: written in C++, every column of the product needs a separate 64-bit addition for the incoming carry, because the
compiler re-associates the carry to the end of the column's v_mad_u64_u32 chain.  In assembly the
chain of column k simply starts from the shifted accumulator of column k-1: 17 (mul), 17 (sqr), 17 (mul2) fewer VALU
instructions per product.  Register use: the 64-bit accumulator is a fixed pair (v[ACC:ACC+1]) because inline asm cannot
name the low half of a 64-bit operand; p's limbs and n0 are SGPR operands (the compiler keeps them resident).

usage: python tools/gen_fq29_asm.py > msm-webgpu_amd/csrc/fq29_asm.h                 (9 limbs of 29 bits: the 254 / 255-bit fields)
       python tools/gen_fq29_asm.py 14 28 254 > msm-webgpu_amd/csrc/fq28x14_asm.h    (14 limbs of 28 bits: BLS12-381; accumulator v[254:255])
"""
import sys

NL = int(sys.argv[1]) if len(sys.argv) > 1 else 9     # limbs
LB = int(sys.argv[2]) if len(sys.argv) > 2 else 29    # bits per limb
MASK = (1 << LB) - 1
ACC = int(sys.argv[3]) if len(sys.argv) > 3 else 166  # v[166:167]: keeps a kernel that uses these blocks within 168 VGPRs (3 waves per SIMD)
acc, acclo = "v[%d:%d]" % (ACC, ACC + 1), "v%d" % ACC


def block(kind, inplace=None):
    """kind: 'mul' (a*b), 'mul2' (a*b + c*d), 'mul4' (a*b + c*d + e*f + g*h: the component of an Fq2 fq_mul2, one reduction for four
    products -- every operand exact, limbs < 2^LB + 64), 'sqr' (a*a; operand list a, a2 = 2a).
    inplace: name of the operand ('a' for mul, 'c' for mul2) whose registers receive the result: result limb j is column 9 + j, and
    limb j of the first factor of a product is last read in column j + 8, so the register is free by then; the m[k] are scratch."""
    L = []
    first = True

    def mad(x, y):
        nonlocal first
        L.append("v_mad_u64_u32 %s, vcc, %s, %s, %s" % (acc, x, y, "0" if first else acc))
        first = False

    for k in range(2 * NL):
        lo, hi = max(0, k - (NL - 1)), min(k, NL - 1)
        if kind in ("mul", "mul2", "mul4"):
            for i in range(lo, hi + 1):
                mad("%%[a%d]" % (k - i), "%%[b%d]" % i)
            if kind in ("mul2", "mul4"):
                for i in range(lo, hi + 1):
                    mad("%%[c%d]" % (k - i), "%%[d%d]" % i)
            if kind == "mul4":
                for x, y in (("e", "f"), ("g", "h")):
                    for i in range(lo, hi + 1):
                        mad("%%[%s%d]" % (x, k - i), "%%[%s%d]" % (y, i))
        else:
            if k % 2 == 0 and k // 2 < NL:
                mad("%%[a%d]" % (k // 2), "%%[a%d]" % (k // 2))
            for i in range(lo, hi + 1):
                if 2 * i < k:
                    mad("%%[a%d]" % i, "%%[t%d]" % (k - i))  # t = 2a
        for i in range(lo, min(k, NL)):
            if 1 <= k - i <= NL - 1:
                mad("%%[m%d]" % i, "%%[p%d]" % (k - i))
        if first:  # column 17 of a product has no term of its own, only the carry (already in acc)
            pass
        if k < NL:
            L.append("v_mul_lo_u32 %%[m%d], %s, %%[n0]" % (k, acclo))
            L.append("v_and_b32 %%[m%d], 0x%x, %%[m%d]" % (k, MASK, k))
            mad("%%[m%d]" % k, "%[p0]")
        elif k < 2 * NL - 1:
            dst = "%%[m%d]" % (k - NL) if not inplace else "%%[%s%d]" % (inplace, k - NL)
            L.append("v_and_b32 %s, 0x%x, %s" % (dst, MASK, acclo))   # result limb k-9 reuses the register of m[k-9] (or of the in-place operand)
        else:
            L.append("v_mov_b32 %s, %s" % ("%%[m%d]" % (NL - 1) if not inplace else "%%[%s%d]" % (inplace, NL - 1), acclo))
        if k < 2 * NL - 1:
            L.append("v_lshrrev_b64 %s, %d, %s" % (acc, LB, acc))
    return L


def emit(name, kind):
    lines = block(kind)
    outs = ", ".join('[m%d] "=&v"(r.v[%d])' % (i, i) for i in range(NL))
    if kind == "mul":
        ins = ", ".join('[a%d] "v"(a.v[%d])' % (i, i) for i in range(NL)) + ", " + ", ".join('[b%d] "v"(b.v[%d])' % (i, i) for i in range(NL))
        sig = "const fq& a, const fq& b"
        pre = ""
    elif kind == "mul2":
        ins = ", ".join('[%s%d] "v"(%s.v[%d])' % (n, i, n if n != "c" else "c_", i) for n in "abcd" for i in range(NL))
        sig = "const fq& a, const fq& b, const fq& c_, const fq& d"
        pre = ""
    elif kind == "mul4":
        ins = ", ".join('[%s%d] "v"(%s.v[%d])' % (n, i, n if n != "c" else "c_", i) for n in "abcdefgh" for i in range(NL))
        sig = "const fq& a, const fq& b, const fq& c_, const fq& d, const fq& e, const fq& f, const fq& g, const fq& h"
        pre = ""
    else:
        ins = ", ".join('[a%d] "v"(a.v[%d])' % (i, i) for i in range(NL)) + ", " + ", ".join('[t%d] "v"(t[%d])' % (i, i) for i in range(1, NL))
        sig = "const fq& a"
        pre = "  uint32_t t[%d];\n#pragma unroll\n  for (int i = 0; i < %d; i++) t[i] = a.v[i] << 1;\n" % (NL, NL)
    ins += ", " + ", ".join('[p%d] "s"(FQ_P29[%d])' % (j, j) for j in range(NL)) + ', [n0] "s"(FQ_N0_29)'
    clob = ['"vcc"', '"v%d"' % ACC, '"v%d"' % (ACC + 1)]
    print("__device__ __forceinline__ fq %s(%s) {" % (name, sig))
    print("  fq r;")
    sys.stdout.write(pre)
    print("  asm(")
    for ln in lines:
        print('      "%s\\n"' % ln)
    print("      : %s" % outs)
    print("      : %s" % ins)
    print("      : %s);" % ", ".join(clob))
    print("  return r;")
    print("}")
    print()


def emit_inplace(name, kind):
    """a = a * b   /   c = a * b + c * d : the result overwrites one operand (no copy back into a loop-carried accumulator)"""
    ip = "a" if kind == "mul" else "c"
    lines = block(kind, ip)
    tmps = ", ".join('[m%d] "=&v"(m[%d])' % (i, i) for i in range(NL))
    if kind == "mul":
        io = ", ".join('[a%d] "+v"(a.v[%d])' % (i, i) for i in range(NL))
        ins = ", ".join('[b%d] "v"(b.v[%d])' % (i, i) for i in range(NL))
        sig = "fq& a, const fq& b"
    else:
        io = ", ".join('[c%d] "+v"(c_.v[%d])' % (i, i) for i in range(NL))
        ins = ", ".join('[%s%d] "v"(%s.v[%d])' % (n, i, n, i) for n in "abd" for i in range(NL))
        sig = "const fq& a, const fq& b, fq& c_, const fq& d"
    ins += ", " + ", ".join('[p%d] "s"(FQ_P29[%d])' % (j, j) for j in range(NL)) + ', [n0] "s"(FQ_N0_29)'
    clob = ['"vcc"', '"v%d"' % ACC, '"v%d"' % (ACC + 1)]
    print("__device__ __forceinline__ void %s(%s) {" % (name, sig))
    print("  uint32_t m[%d];" % NL)
    print("  asm(")
    for ln in lines:
        print('      "%s\\n"' % ln)
    print("      : %s, %s" % (io, tmps))
    print("      : %s" % ins)
    print("      : %s);" % ", ".join(clob))
    print("}")
    print()


print("// GENERATED by tools/gen_fq29_asm.py%s -- do not edit.  gfx950 inline-assembly forms of fq_mul / fq_sqr (see that script)." % ("" if NL == 9 else " %d %d %d" % (NL, LB, ACC)))
print("// (no include guard: included by fq29.h once per curve unit, inside the unit's field namespace; the blocks take p's limbs and n0")
print("//  as operands from that namespace's constants)")
print("namespace MSM_FIELD_NS {")
emit("fq_mul_asm", "mul")
emit("fq_sqr_asm", "sqr")
emit("fq_mul2_asm", "mul2")
# 4 x NL x (2^LB + 64)^2 product terms + NL x 2^(2 LB) reduction terms per column must fit the 64-bit accumulator
assert 4 * NL * ((1 << LB) + 64) ** 2 + NL * (1 << (2 * LB)) < (1 << 64), "mul4 column overflow"
emit("fq_mul4_asm", "mul4")
emit_inplace("fq_mul_ip_asm", "mul")
emit_inplace("fq_mul2_ip_asm", "mul2")
print("}  // namespace MSM_FIELD_NS")
