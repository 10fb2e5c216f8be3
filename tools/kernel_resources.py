#!/usr/bin/env python3
"""Registers / scratch / LDS of every kernel in the device assembly the build left behind (build/temps*/…gfx950.s).
usage: kernel_resources.py [temps dir] [other temps dir]   (two directories: only the kernels that differ are listed)"""
import glob
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = ("next_free_vgpr", "accum_offset", "private_segment_fixed_size", "group_segment_fixed_size")


def resources(temps):
    out = {}
    for path in sorted(glob.glob(os.path.join(temps, "*-hip-amdgcn-amd-amdhsa-gfx950.s"))):
        name = None
        for ln in open(path):
            m = re.match(r"\s*\.amdhsa_kernel (\S+)", ln)
            if m:
                name = m.group(1)
                out[name] = {}
            m = re.match(r"\s*\.amdhsa_(\w+)\s+(\S+)", ln)
            if m and name and m.group(1) in KEYS:
                out[name][m.group(1)] = int(m.group(2))
    return out


def fmt(d):
    return "vgpr %3d agpr %3d scratch %4d lds %6d" % (min(d["next_free_vgpr"], d["accum_offset"]), max(0, d["next_free_vgpr"] - d["accum_offset"]),
                                                        d["private_segment_fixed_size"], d["group_segment_fixed_size"])


if __name__ == "__main__":
    a = resources(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "build", "temps"))
    if len(sys.argv) > 2:
        b = resources(sys.argv[2])
        for k in sorted(a):
            if k in b and a[k] != b[k]:
                print("%-90s %s | %s" % (k[:90], fmt(a[k]), fmt(b[k])))
    else:
        for k in sorted(a):
            print("%-90s %s" % (k[:90], fmt(a[k])))
