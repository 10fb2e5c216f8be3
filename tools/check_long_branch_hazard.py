#!/usr/bin/env python3
"""Static check of the gfx950 device code for a code-generation hazard that hipcc (ROCm 7.2) can produce in LARGE kernels.

Background ("the fault of the all-assembly build").  A kernel larger than the +-128 KiB reach of
s_branch / s_cbranch gets its far branches expanded AFTER wait-count insertion into
        s_getpc_b64 s[a:a+1] ; s_add_u32 ; s_addc_u32 ; s_setpc_b64 s[a:a+1]
with the SGPR pair taken from whatever is dead at that point.  "Dead" ignores scalar loads that are still IN FLIGHT: if an
s_load_* into s[a:a+1] was issued earlier on that path and no `s_waitcnt lgkmcnt(0)` lies in between, the load returns after
s_getpc_b64 wrote the pair and overwrites it -- the wave then jumps to (loaded value + offset), i.e. it fetches instructions
from a data pointer: "Memory access fault by GPU", intermittently (it depends on whether the load or the s_getpc wins).
This is what made the diagnostic build with the inline-assembly multipliers in every kernel fault in k_bpr_w256.

The check: per function, a forward data-flow over the basic blocks of the compiler's assembly output tracks the SGPRs with a
scalar load possibly in flight and reports every s_getpc_b64 whose destination pair is among them.

A second check of the same expansions (round 3): the pair is taken from what the expansion believes to be dead -- a backward
liveness analysis over the SGPRs (definitions and uses of every instruction, inline assembly included; calls use the argument
registers) reports every expanded branch whose pair is LIVE at the branch target, i.e. read there before it is written.  No build
of this repository has shown one; the check is cheap insurance for the same family of expansions.

usage: check_long_branch_hazard.py [file.s]        (no argument: compiles msm-webgpu_amd/csrc/msm_hip.hip to assembly first)
exit status 0 = clean, 1 = hazard found.
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def sgprs(tok):
    m = re.match(r"s\[(\d+):(\d+)\]$", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"s(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def functions(text):
    cur, name = None, None
    for ln in text.split("\n"):
        m = re.match(r"^([A-Za-z_][\w$.]*):\s*; @", ln)
        if m:
            name, cur = m.group(1), []
            continue
        if cur is not None:
            if ln.startswith(".Lfunc_end"):
                yield name, cur
                cur = None
            else:
                cur.append(ln)


def analyse(name, lines):
    # instruction list with labels
    instrs, labels = [], {}
    for ln in lines:
        t = ln.split(";")[0].strip()
        if not t:
            continue
        m = re.match(r"^(\.L[\w$]+):$", t)
        if m:
            labels[m.group(1)] = len(instrs)
            continue
        if t.startswith("."):
            continue
        parts = t.split(None, 1)
        ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
        instrs.append((parts[0], ops, t))
    n = len(instrs)
    if n == 0:
        return []
    # successors
    succ = [[] for _ in range(n)]
    pending_target = None  # label of the long jump being assembled (s_add_u32 sX, sX, (.LBBn_m-.Lpost_getpcK)&...)
    for i, (op, ops, t) in enumerate(instrs):
        m = re.search(r"\((\.LBB\d+_\d+)-\.Lpost_getpc\d+\)", t)
        if m:
            pending_target = m.group(1)
        if op == "s_branch":
            succ[i] = [labels[ops[0]]]
        elif op.startswith("s_cbranch"):
            succ[i] = [labels[ops[-1]]] + ([i + 1] if i + 1 < n else [])
        elif op == "s_setpc_b64":
            succ[i] = [labels[pending_target]] if pending_target in labels and ops[0] != "s[30:31]" else []
            pending_target = None
        elif op in ("s_endpgm", "s_trap"):
            succ[i] = []
        else:
            succ[i] = [i + 1] if i + 1 < n else []
    # forward data-flow: SGPRs with a scalar load possibly in flight on entry to instruction i
    state = [None] * n
    state[0] = frozenset()
    work = [0]
    hazards = {}
    while work:
        i = work.pop()
        cur = set(state[i])
        op, ops, t = instrs[i]
        if op == "s_getpc_b64" and cur & sgprs(ops[0]):
            hazards[i] = (t, sorted(cur & sgprs(ops[0])))
        if op.startswith("s_load_") or op.startswith("s_buffer_load_") or op.startswith("s_scratch_load"):
            cur |= sgprs(ops[0])
        elif op == "s_waitcnt" and re.search(r"lgkmcnt\(0\)", t):
            cur = set()  # scalar loads return out of order: only lgkmcnt(0) settles them
        elif op == "s_waitcnt" and not re.search(r"[a-z]", " ".join(ops)):  # raw immediate form: lgkmcnt field = bits 11:8
            if (int(ops[0], 0) >> 8) & 0xF == 0:
                cur = set()
        elif op == "s_swappc_b64":
            cur = set()  # every callee starts with s_waitcnt vmcnt(0) expcnt(0) lgkmcnt(0)
        out = frozenset(cur)
        for j in succ[i]:
            merged = out if state[j] is None else state[j] | out
            if merged != state[j]:
                state[j] = merged
                work.append(j)
    return [(name, t, regs) for (t, regs) in hazards.values()]


NO_DST = ("s_cmp", "s_bitcmp", "s_cbranch", "s_setpc", "s_waitcnt", "s_nop", "s_endpgm", "s_barrier", "s_sleep", "s_sendmsg", "s_setprio",
          "s_inst_prefetch", "s_setreg", "s_branch", "s_trap", "s_icache", "s_dcache", "s_ttrace", "s_store", "s_buffer_store", "s_scratch_store")
CARRY = ("v_add_co_u32", "v_sub_co_u32", "v_subrev_co_u32", "v_addc_co_u32", "v_subb_co_u32", "v_subbrev_co_u32", "v_mad_u64_u32", "v_mad_i64_i32",
         "v_div_scale")


def sgpr_defs_uses(op, ops):
    """SGPRs written / read by one instruction.  When in doubt a register counts as READ and not as written (a false alarm, never a miss)."""
    d, u = set(), set()
    if op.startswith("s_"):
        if op.startswith(NO_DST):
            for o in ops:
                u |= sgprs(o)
        elif op == "s_swappc_b64":
            d |= sgprs(ops[0])
            u |= sgprs(ops[1]) | set(range(0, 36))  # the callee's arguments
        else:
            if ops:
                d |= sgprs(ops[0])
            for o in ops[1:]:
                u |= sgprs(o)
            if op.startswith("s_cmov"):
                u |= sgprs(ops[0])
    elif op.startswith("v_"):
        base = op.replace("_e64", "").replace("_e32", "").replace("_sdwa", "").replace("_dpp", "")
        if base.startswith("v_cmp") and not base.startswith("v_cmpx") and op.endswith("_e64") and ops:
            d |= sgprs(ops[0])
            for o in ops[1:]:
                u |= sgprs(o)
        elif base in ("v_readfirstlane_b32", "v_readlane_b32"):
            d |= sgprs(ops[0])
            for o in ops[1:]:
                u |= sgprs(o)
        elif base.startswith(CARRY):
            if len(ops) > 1:
                d |= sgprs(ops[1])
            for o in ops[2:]:
                u |= sgprs(o)
        else:
            for o in ops:
                u |= sgprs(o)
    else:
        for o in ops:
            for t in re.findall(r"s\[\d+:\d+\]|s\d+", o):
                u |= sgprs(t)
    return d, u


def analyse_liveness(name, lines):
    """expanded long branches of one function whose scratch pair is live at the branch target"""
    if not any("s_getpc_b64" in ln for ln in lines):
        return []
    instrs, labels = [], {}
    for ln in lines:
        t = ln.split(";")[0].strip()
        if not t:
            continue
        m = re.match(r"^(\.L[\w$]+):$", t)
        if m:
            labels[m.group(1)] = len(instrs)
            continue
        if t.startswith("."):
            continue
        parts = t.split(None, 1)
        instrs.append((parts[0], [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else [], t))
    n = len(instrs)
    succ = [[] for _ in range(n)]
    pending, getpc_at, jumps = None, None, []
    for i, (op, ops, t) in enumerate(instrs):
        m = re.search(r"\((\.LBB\d+_\d+)-\.Lpost_getpc\d+\)", t)
        if m:
            pending = m.group(1)
        if op == "s_getpc_b64":
            getpc_at = i
        if op == "s_branch":
            succ[i] = [labels[ops[0]]]
        elif op.startswith("s_cbranch"):
            succ[i] = [labels[ops[-1]]] + ([i + 1] if i + 1 < n else [])
        elif op == "s_setpc_b64":
            if pending in labels and ops[0] != "s[30:31]":
                succ[i] = [labels[pending]]
                jumps.append((getpc_at, sgprs(ops[0]), labels[pending], pending))
            pending = None
        elif op not in ("s_endpgm", "s_trap"):
            succ[i] = [i + 1] if i + 1 < n else []
    du = [sgpr_defs_uses(op, ops) for op, ops, _ in instrs]
    live_in = [set() for _ in range(n)]
    changed = True
    while changed:
        changed = False
        for i in range(n - 1, -1, -1):
            out = set()
            for j in succ[i]:
                out |= live_in[j]
            new = du[i][1] | (out - du[i][0])
            if new != live_in[i]:
                live_in[i] = new
                changed = True
    return [(name, instrs[gi][2], lab, sorted(pair & live_in[ti])) for gi, pair, ti, lab in jumps if pair & live_in[ti]]


def compile_to_asm(extra):
    """Device assembly of the product, one file per translation unit: the files the library's own build left behind when they are
    current (msm-webgpu_amd/build.py keeps the compiler's intermediate files), else fresh -S compiles (minutes)."""
    sys.path.insert(0, os.path.join(ROOT, "msm-webgpu_amd"))
    import build as _b

    if not extra and _b.device_asm_is_current():
        return _b.device_asm_files()
    outdir = tempfile.mkdtemp(prefix="msm_hip_asm_")
    outs = []
    for unit in _b.TRANSLATION_UNITS:
        out = os.path.join(outdir, os.path.splitext(unit)[0] + ".s")
        cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S",
               os.path.join(ROOT, "msm-webgpu_amd", "csrc", unit), "-o", out] + extra
        subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
        outs.append(out)
    return outs


def check_file(path):
    found = []
    text = open(path).read()
    long_branches = text.count(".Lpost_getpc") // 3
    live = []
    for name, lines in functions(text):
        found += analyse(name, lines)
        live += analyse_liveness(name, lines)
    return long_branches, found, live


def main():
    args = sys.argv[1:]
    paths = [a for a in args if a.endswith(".s")] or compile_to_asm(args)
    bad = 0
    for path in paths:
        long_branches, found, live = check_file(path)
        print("%s: %d expanded long branches, %d with a scalar load in flight into their register pair, %d with a live pair" % (path, long_branches, len(found), len(live)))
        for name, t, regs in found:
            print("  HAZARD in %s: `%s` while s_load into s%s may be in flight" % (name, t, regs))
        for name, t, lab, regs in live:
            print("  HAZARD in %s: `%s` (long branch to %s) overwrites s%s, which is read at the target before it is written" % (name, t, lab, regs))
        bad += len(found) + len(live)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
