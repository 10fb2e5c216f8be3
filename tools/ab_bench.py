"""A/B of two builds on the SAME GPU box (box-to-box variation is larger than most kernel tweaks): the in-tree libmsm_hip.so
("new") against msm-webgpu_amd/_variants/base.so ("base", e.g. a copy of the previous build), three alternating runs of bench.py.
usage: python tools/ab_bench.py [bench.py args]"""
import os, sys, subprocess, json
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
main = os.path.join(root, "msm-webgpu_amd", "libmsm_hip.so")
alt = os.path.join(root, "msm-webgpu_amd", "_variants", "base.so")
keep = main + ".keep"
os.rename(main, keep)
args = sys.argv[1:] or ["--steps", "60", "--warmup", "8"]
try:
    for rnd in range(3):
        for name, so in (("new", keep), ("base", alt)):
            if os.path.lexists(main): os.remove(main)
            os.symlink(so, main)
            out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--no-cpu-baseline"] + args, capture_output=True, text=True).stdout
            d = json.loads(out.strip().splitlines()[-1])
            print(rnd, name, round(d["value"], 1), round(d["ms_per_step"], 3), "smvp", round(d["smvp_ms_pipelined"], 3), flush=True)
finally:
    if os.path.lexists(main): os.remove(main)
    os.rename(keep, main)
