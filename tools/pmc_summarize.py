#!/usr/bin/env python3
"""Summarise a tools/profile_round.sh run: per-kernel averages of FETCH_SIZE / WRITE_SIZE (KiB, as rocprofv3 reports them),
the calibration factors measured on known byte counts in this engine's access patterns, and the corrected HBM traffic of
k_smvp_chunks per launch.  Usage: python tools/pmc_summarize.py gpurun_out/prof_<tag> <logn> <w_local> [out.json] [bases]
(bases = plain | endomorphism: with the endomorphism a launch has w_local = 8 bucket sets of 2 * 2^logn entries)"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def per_kernel(dirpath, counter):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(dirpath, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row["Counter_Name"] == counter:
                    name = row["Kernel_Name"].split("(")[0]
                    acc[name].append(float(row["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


def kernel_stats(dirpath):
    out = {}
    for f in glob.glob(os.path.join(dirpath, "**", "*kernel_stats.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                out[row["Name"].split("(")[0]] = {"calls": int(row["Calls"]), "avg_ns": float(row["AverageNs"])}
    return out


def main():
    root, logn, w_local = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    out_json = sys.argv[4] if len(sys.argv) > 4 else None
    bases = sys.argv[5] if len(sys.argv) > 5 else "plain"
    fetch, nfetch = per_kernel(os.path.join(root, "pmc_fetch"), "FETCH_SIZE")
    write, _ = per_kernel(os.path.join(root, "pmc_write"), "WRITE_SIZE")
    cfetch, _ = per_kernel(os.path.join(root, "cal_fetch"), "FETCH_SIZE")
    cwrite, _ = per_kernel(os.path.join(root, "cal_write"), "WRITE_SIZE")
    stats = kernel_stats(os.path.join(root, "trace"))
    GiB = 1 << 30
    cal = {
        "stream_read_16B_per_lane": cfetch.get("k_stream_read", 0) * 1024 / GiB,
        "gather_64B_records": cfetch.get("k_gather64", 0) * 1024 / GiB,
        "stream_write_16B_per_lane": cwrite.get("k_stream_write", 0) * 1024 / GiB,
        "scatter_160B_records": cwrite.get("k_scatter160", 0) * 1024 / GiB,
    }
    print("calibration: reported bytes / true bytes (1 GiB moved by each kernel)")
    for k, v in cal.items():
        print("  %-28s %.3f" % (k, v))
    print("\nper-kernel averages (KiB as reported, uncorrected):")
    for k in sorted(set(fetch) | set(write)):
        st = stats.get(k, stats.get("msmk::" + k, {}))
        print("  %-28s FETCH_SIZE %12.1f  WRITE_SIZE %12.1f  avg_us %10.1f  calls %s" % (
            k.replace("msmk::", ""), fetch.get(k, 0), write.get(k, 0), st.get("avg_ns", 0) / 1e3, st.get("calls", "")))
    name = "msmk::k_smvp_chunks"
    n = (2 if bases == "endomorphism" else 1) << logn
    alg = n * w_local * 68 + w_local * 32768 * 96
    f_raw, w_raw = fetch.get(name, 0) * 1024, write.get(name, 0) * 1024
    f_corr = f_raw / cal["gather_64B_records"] if cal["gather_64B_records"] else None
    w_corr = w_raw / cal["scatter_160B_records"] if cal["scatter_160B_records"] else None
    res = {"logn": logn, "w_local": w_local, "bases": bases, "kernel": "k_smvp_chunks", "algorithmic_bytes": alg,
           "fetch_bytes_reported": f_raw, "write_bytes_reported": w_raw,
           "fetch_calibration_gather64": cal["gather_64B_records"], "write_calibration_scatter160": cal["scatter_160B_records"],
           "fetch_bytes_corrected": f_corr, "write_bytes_corrected": w_corr,
           "hbm_bytes_per_launch": (f_corr + w_corr) if f_corr is not None and w_corr is not None else None,
           "kernel_avg_us_rocprof": stats.get(name, {}).get("avg_ns", 0) / 1e3,
           "note": "FETCH_SIZE / WRITE_SIZE are fabric-side L2 request counters; Infinity-Cache hits are counted too"}
    print("\n" + json.dumps(res, indent=1))
    if out_json:
        with open(out_json, "w") as fh:
            json.dump(res, fh, indent=1)


if __name__ == "__main__":
    main()
