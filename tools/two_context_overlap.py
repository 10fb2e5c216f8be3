"""Experiment: can the sort of one MSM run under the SMVP of another?  Two engine contexts on one GPU, their launches interleaved
(each context keeps its own two launches in flight), against one context alone.  usage: two_context_overlap.py [logn [timed MSMs]] (MSM_HIP_SO selects the build)"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # as bench.py: the engine streams of both contexts on their own hardware queues
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import msm_webgpu_amd as m
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 20
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 120
n = 1 << logn
ctxs = [m.MsmContext(0), m.MsmContext(0)]
pts = ctxs[0].sample_points(n, 1)
sc = [ctxs[0].sample_scalars(n, 2 + i) for i in range(2)]
for c in ctxs:
    c.set_bases(pts, endomorphism=True)
    c.set_stage_timing(0)
def run(cs, k, depth=2):
    q = {id(c): [] for c in cs}
    cnt = {id(c): 0 for c in cs}
    for j in range(k):
        c = cs[j % len(cs)]
        slot = cnt[id(c)] % depth
        cnt[id(c)] += 1
        c.launch(sc[j & 1], slot)
        q[id(c)].append(slot)
        if len(q[id(c)]) == depth:
            c.finish(q[id(c)].pop(0))
    for c in cs:
        for s0 in q[id(c)]:
            c.finish(s0)
for cs, name in ((ctxs[:1], "one context"), (ctxs, "two contexts interleaved")):
    run(cs, max(4, steps // 6))
    torch.cuda.synchronize(); t0 = time.perf_counter(); run(cs, steps); torch.cuda.synchronize()
    print("2^%d %s [%s]: %.4f ms per MSM" % (logn, name, os.path.basename(os.environ.get("MSM_HIP_SO", "product")), (time.perf_counter() - t0) / steps * 1e3), flush=True)
