"""Does the per-MSM time drift over the first steps of a run (clock ramp)?  bench.py's pipeline (depth 2, endomorphism bases), one
time stamp per finished MSM.  usage: step_time_trend.py [steps] [warmup]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import msm_webgpu_amd as m
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
warm = int(sys.argv[2]) if len(sys.argv) > 2 else 5
pre = sys.argv[3] if len(sys.argv) > 3 else ""   # "sampler": ~25 ms of unrelated GPU work right before the warm-up; "sleep": 50 ms idle after it
n = 1 << 20
ctx = m.MsmContext(0)
pts = ctx.sample_points(n, 1)
sc = [ctx.sample_scalars(n, 2 + i) for i in range(2)]
ctx.set_bases(pts, endomorphism=True)
ctx.set_stage_timing(1)
def run(k, stamps=None):
    fl = []
    for j in range(k):
        ctx.launch(sc[j & 1], j % 2)
        fl.append(j % 2)
        if len(fl) == 2:
            ctx.finish(fl.pop(0))
            if stamps is not None:
                stamps.append(time.perf_counter())
    for s in fl:
        ctx.finish(s)
        if stamps is not None:
            stamps.append(time.perf_counter())
if pre == "sampler":
    ctx.sample_points(1 << 20, 99)
run(warm)
torch.cuda.synchronize()
if pre == "sleep":
    time.sleep(0.05)
st = [time.perf_counter()]
run(steps, st)
torch.cuda.synchronize()
d = [(st[i + 1] - st[i]) * 1e3 for i in range(len(st) - 1)]
for i in range(0, len(d), 10):
    print("MSMs %3d..%3d: %s" % (i, i + 9, " ".join("%.3f" % x for x in d[i:i + 10])))
print("steps %d warmup %d pre=%r" % (steps, warm, pre))
print("mean of first 20: %.4f   mean of 21..: %.4f   total/steps %.4f" % (sum(d[:20]) / 20, sum(d[20:]) / max(1, len(d) - 20), (st[-1] - st[0]) * 1e3 / steps))
