#!/usr/bin/env python3
"""Benchmark of the hot path: BN254 G1 MSM at 2^20 points, 16-bit signed-bucket windows (BASELINE.json configs[1]).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--logn 20] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A step = one whole MSM (decompose, sort, SMVP accumulate, bucket reduce on the GPU; window combine on the host) over
synthetic inputs that are already resident in HBM when the timed region starts.  N = 1: one GPU does all 16 windows.
N > 1: the 16 Pippenger windows of every MSM are sharded over the ranks (one process per GPU); a rank puts its shares of
16 / windows_per_rank consecutive MSMs through one launch (config.msms_per_launch; one MSM's share cannot fill a GPU),
ONE RCCL all-gather brings their window sums to every rank, host combine -- total work is fixed, so "scaling" is
"strong".  Rank 0 prints ONE JSON line.

roofline: the SMVP accumulate kernel.  achieved = ALGORITHMIC bytes per launch (N * W_local * 68 B read +
W_local * 2^15 * 96 B written) / its average duration measured with HIP events on the engine's own stream over the
timed steps; peak = 8000 GB/s (HBM3E).  traffic (PMC-measured HBM bytes per launch) is read from
profiles/smvp_pmc_traffic.json when that file matches the workload, else null.
cpu_baseline: the CPU oracle's restatement of halo2curves' serial msm (kind "port", not halo2curves itself) timed on
this host on the same inputs, rank 0, N = 1 only; it is also the bit-exact check of the GPU result.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# the engine runs 3 HIP streams and the sharded path adds torch's copy stream and RCCL's: keep them on separate hardware
# queues (ROCm's default is 4 per process; must be set before the HIP runtime initialises)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

HBM_PEAK_GBS = 8000.0
# the SMVP accumulate is VALU-issue bound: one mixed addition = 1467 v_mad_u64_u32 per lane; measured peak issue
# rate of that instruction = 1024 SIMDs x 64 lanes / 2.3 ns (profiles/r01_ubench_valu_rates.txt)
MADS_PER_MIXED_ADD = 1467
VALU_PEAK_LANE_MADS = 1024 * 64 / 2.3e-9
NUM_WINDOWS = 16
BUCKETS = 1 << 15


def smvp_algorithmic_bytes(n, w_local, buckets=BUCKETS):
    """n entries per window at most (one per point the recode reads), w_local bucket sets of `buckets` buckets"""
    return n * w_local * (64 + 4) + w_local * buckets * 96


def smvp_shape(m, ctx, n, bases_mode, bits):
    """(inputs per bucket set, bucket sets, buckets per set) of one whole MSM's SMVP launch over n points at a window size"""
    nwin, buckets = m.MsmContext.window_config(bits)
    if bases_mode == "endomorphism":
        return 2 * n, m.MsmContext.endomorphism_window_count(bits), buckets
    if bases_mode == "tables":
        return NUM_WINDOWS * n, 1, buckets
    if bases_mode == "tables_wide":
        wb = ctx.wide_bits()
        return ((254 + wb) // wb) * n / (1 << (wb - 16)), 1 << (wb - 16), buckets
    return n, nwin, buckets


class WholeMsmRunner:
    """Back-to-back whole MSMs of n points on one GPU through the launch / finish halves of the C ABI: a launch holds one MSM (n >= 2^19)
    or up to `group` whole small MSMs (as msm_hip_run_batch_* does; vector k of a launch is scalar set k & 1); `depth` launches in flight
    over the engine's result slots, so that the host window combines (47 us per MSM) and the launch calls of one launch run under the
    device work of the others.  Create it AFTER set_bases: the grouping follows the base mode."""

    def __init__(self, m, torch, ctx, n, scalar_sets, bases_mode, depth=0):
        self.m, self.ctx, self.n, self.sets, self.bases_mode = m, ctx, n, scalar_sets, bases_mode
        self.group = max(1, min(ctx.batch_group_size(n), 8))
        self.scalars = torch.cat([scalar_sets[k & 1] for k in range(self.group)], dim=0).contiguous() if self.group > 1 else None
        # launches in flight (4 result slots): 2 where one launch is one large MSM, 3 for grouped small MSMs, whose host combines (4 per launch)
        # would otherwise sit between launches (measured: +5 % at 2^16, +3 % at 2^18, nothing at 2^20)
        self.depth = max(1, min(4, depth or int(os.environ.get("BENCH_PIPE_DEPTH", "0")) or (3 if self.group > 1 else 2)))

    def sizes(self, count):
        """`count` MSMs in ceil(count / group) launches: full launches and the remainder LAST when it is at least half a launch -- what
        follows the last launch's main-stream work (bucket reduce, host combines) is not hidden by a next launch and is shorter for a
        smaller one (measured at 2^16, 20 MSMs: 0.293 vs 0.328 ms per MSM) --, else launches of nearly equal size."""
        g = self.group
        k = -(-count // g)
        rem = count % g
        if rem and 2 * rem >= g:
            return [g] * (count // g) + [rem]
        return [count // k + (1 if i < count % k else 0) for i in range(k)]

    def last_set(self, count):
        """index of the scalar set the last MSM of run(count) uses"""
        return (count - 1) & 1 if self.group == 1 else (self.sizes(count)[-1] - 1) & 1

    def run(self, count, stages=None):
        """`count` MSMs; returns the last result.  `stages` (a list) receives (smvp kernel ms, bucket sets, window bits) of every launch
        (needs stage timing level >= 1)."""
        ctx, n, result, pending = self.ctx, self.n, None, []

        def collect():
            slot0, gs0 = pending.pop(0)
            res = ctx.finish(slot0) if self.group == 1 else ctx.finish_batch(slot0, gs0)[-1]
            if stages is not None:
                bits = ctx.last_window_bits()
                stages.append((ctx.stage_ms()["smvp"], gs0 * smvp_shape(self.m, ctx, n, self.bases_mode, bits)[1], bits))
            return res

        for k, gs in enumerate(self.sizes(count)):
            slot = k % self.depth
            if self.group == 1:
                ctx.launch(self.sets[k & 1], slot)
            else:
                ctx.launch_batch(self.scalars[: gs * n], n, slot)
            pending.append((slot, gs))
            if len(pending) == self.depth:
                result = collect()
        while pending:
            result = collect()
        return result

    def roofline(self, stages):
        """the SMVP accumulate kernel over the launches in `stages`: algorithmic bytes per launch / average duration vs HBM peak"""
        if not stages:
            return None
        ms = sum(s[0] for s in stages) / len(stages)
        alg = sum(smvp_algorithmic_bytes(smvp_shape(self.m, self.ctx, self.n, self.bases_mode, b)[0], w, 1 << (b - 1)) for _, w, b in stages) / len(stages)
        ach = alg / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        return {"bound": "hbm", "kernel": "k_smvp_chunks", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                "algorithmic_bytes": alg, "kernel_ms": ms, "launches": len(stages)}


def measure_configs(m, torch, ctx, args, spec, points_main, sets_main):
    """BASELINE.json's configs other than the headline's, each with its own timed region on this one GPU (inputs resident in HBM), the SMVP
    kernel's roofline fraction from its own HIP events, and a check of its result.  Returns (configs, checks): `checks` are the comparisons
    with the CPU oracle, run later in the cpu_baseline leg (the oracle is never imported on the measurement path).
      c1  2^16 MSM (BASELINE: the reference's CPU-runnable case): GPU throughput and latency; the oracle's own time is its CPU figure
      c3  2^20, windows over 8 GPUs: ONE rank's share (8 MSMs' shares per launch) timed on this GPU -- a projection, no collective --, and all
          8 ranks' shares run one after the other, gathered in rank order and combined == the whole MSM
      c4  2^24 MSM: >= 5 timed MSMs; whole == the 8 plain window-range shares combined; a 2^16 slice against the oracle
      c5  64 x 2^18 over one shared base through msm_hip_run_batch_device, default (endomorphism) and wide-table bases; one vector
          against the oracle"""
    from msm_webgpu_amd.sharding import ShardedMsmPipeline, gathered_window_sums, window_range

    spec = dict(item.split(":") for item in spec.split(",") if item)
    cfg, checks = {}, []
    W = max(args.warmup, 1)
    host = lambda t: t.cpu().numpy().tobytes()

    def sync():
        torch.cuda.synchronize()

    def timed_run(runner, warm, count):
        ctx.set_stage_timing(1)
        runner.run(warm, None)
        st = []
        sync()
        t0 = time.perf_counter()
        last = runner.run(count, st)
        sync()
        return time.perf_counter() - t0, last, st

    if "c1" in spec:
        logn = int(spec["c1"])
        n = 1 << logn
        pts = ctx.sample_points(n, 0xC10001)
        sets = [ctx.sample_scalars(n, 0xC10100 + k) for k in range(2)]
        ctx.set_bases(pts, endomorphism=True)
        r = WholeMsmRunner(m, torch, ctx, n, sets, "endomorphism")
        warm, count = r.group * r.depth + W, max(args.steps, 8 * r.group)
        el, last, st = timed_run(r, warm, count)
        ctx.set_stage_timing(0)
        lat = []
        for i in range(7):
            sync()
            t1 = time.perf_counter()
            ctx.msm(sets[i & 1])
            lat.append((time.perf_counter() - t1) * 1e3)
        cfg["c1"] = {"workload": "2^%d BN254 G1 MSM, one GPU, bases with their endomorphism images, %d whole MSMs per launch" % (logn, r.group),
                     "value": count / el, "unit": "MSM/s", "ms_per_msm": el * 1e3 / count, "timed_msms": count, "untimed_msms_directly_before": warm,
                     "window_bits": st[0][2], "latency_ms_single_msm": sorted(lat)[len(lat) // 2], "roofline": r.roofline(st)}
        checks.append({"config": "c1", "key": "verified_bit_exact_vs_cpu", "points": host(pts), "scalars": host(sets[r.last_set(count)]),
                       "got": last.to_affine_bytes(), "threads": 1, "time_as": "cpu_path"})
        del pts, sets, r

    if "c3" in spec:
        logn = int(spec["c3"])
        n = 1 << logn
        pts = points_main if points_main is not None else ctx.sample_points(n, 0xC30001)
        sets = sets_main if points_main is not None else [ctx.sample_scalars(n, 0xC30100 + k) for k in range(2)]
        world, c3 = 8, {"workload": "2^%d BN254 G1 MSM, windows over 8 GPUs: ONE rank's share timed on this GPU (8 MSMs' shares per launch)" % logn,
                        "note": "a one-GPU projection -- no collective, no second GPU; the driver's SCALE run is the measurement"}
        ctx.set_stage_timing(0)
        for mode in [x for x in os.environ.get("BENCH_C3_MODES", "plain,tables_wide").split(",") if x]:
            wide = mode == "tables_wide"
            if wide:
                ctx.set_wide_bits(int(os.environ.get("BENCH_WIDE_BITS", "19")))
                ctx.set_bases(pts, precompute="wide")
                nwin = 1 << (ctx.wide_bits() - 16)
            else:
                ctx.set_bases(pts)
                nwin = NUM_WINDOWS
            g = max(1, nwin // -(-nwin // world))  # as many MSMs' shares per launch as make up one MSM's worth of bucket sets
            batch = torch.cat([sets[k & 1] for k in range(g)], dim=0).contiguous()
            # (i) every rank's launch, one after the other; what the all-gather would deliver, combined == the whole MSM (plain bases, 16 windows)
            per = -(-nwin // world)
            rec = 2 if wide else 1  # wide shares: (window sum, plain total) pairs
            gathered = torch.zeros((world, g * per * rec, ctx.jb), dtype=torch.uint8, device=batch.device)
            for rk in range(world):
                b, e = window_range(rk, world, nwin)
                if e > b:
                    (ctx.launch_vwindows_batch if wide else ctx.launch_windows_batch)(batch, n, b, e, rk % 3, gathered[rk][: g * (e - b) * rec])
                    ctx.slot_sync(rk % 3)
            hostg = gathered.cpu().numpy()
            if wide:
                got = m.MsmContext.combine_vwindows_batch(gathered_window_sums(hostg.reshape(world, g * per, rec * ctx.jb), g, world, nwin), nwin)
                ctx.set_wide_bits(0)
                ctx.set_bases(pts)
            else:
                got = m.MsmContext.combine_windows_batch(gathered_window_sums(hostg, g, world, nwin), nwin)
            whole = [ctx.msm(sets[k]) for k in range(2)]
            ok = all(bool(got[v] == whole[v & 1]) for v in range(g))
            if wide:
                ctx.set_wide_bits(int(os.environ.get("BENCH_WIDE_BITS", "19")))
                ctx.set_bases(pts, precompute="wide")
            # (ii) rank 0's share through the sharded pipeline, timed
            pipe = ShardedMsmPipeline(ctx, 0, 1, depth=3, msms_per_issue=g, emulate_world=world, wide=wide)
            ctx.set_stage_timing(1)

            def run(count, st=None):
                inflight = []
                for _ in range(-(-count // g)):
                    pipe.issue(batch, n, inputs_complete=True)
                    inflight.append(1)
                    if len(inflight) == pipe.depth:
                        pipe.complete()
                        inflight.pop()
                        if st is not None:
                            st.append(ctx.stage_ms()["smvp"])
                while inflight:
                    pipe.complete()
                    inflight.pop()
                    if st is not None:
                        st.append(ctx.stage_ms()["smvp"])
                return -(-count // g) * g

            warm = run(3 * g + W)
            st = []
            sync()
            t0 = time.perf_counter()
            count = run(max(args.steps, 3 * g), st)
            sync()
            el = time.perf_counter() - t0
            wl = pipe.w_end - pipe.w_begin
            per_set = ((254 + ctx.wide_bits()) // ctx.wide_bits()) * n / nwin if wide else n
            alg = smvp_algorithmic_bytes(per_set, g * wl)
            kms = sum(st) / len(st)
            c3[mode] = {"windows_shared": "%d virtual windows of the wide tables (%d-bit digits, %d additions per point)" % (nwin, ctx.wide_bits(), (254 + ctx.wide_bits()) // ctx.wide_bits()) if wide
                        else "16 windows of 16 bits", "windows_per_rank": wl, "msms_per_launch": g,
                        "ms_per_msm_one_rank_share": el * 1e3 / count, "projected_msm_per_s_at_8_gpus": count / el, "timed_msm_shares": count,
                        "untimed_msm_shares_directly_before": warm, "all_8_shares_combined_equal_whole_msm": ok,
                        "roofline": {"bound": "hbm", "kernel": "k_smvp_chunks", "kernel_ms": kms, "algorithmic_bytes": alg, "achieved": alg / (kms * 1e-3) / 1e9,
                                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS}}
            del pipe, batch, gathered
        ctx.set_wide_bits(0)
        ctx.set_bases(pts)
        ctx.set_stage_timing(0)
        w0 = ctx.msm(sets[0])
        cfg["c3"] = c3
        checks.append({"config": "c3", "key": "whole_msm_verified_bit_exact_vs_cpu", "points": host(pts), "scalars": host(sets[0]), "got": w0.to_affine_bytes()})
        del pts, sets

    if "c4" in spec:
        logn = int(spec["c4"])
        n = 1 << logn
        pts = ctx.sample_points(n, 0xC40001)
        sets = [ctx.sample_scalars(n, 0xC40002 + k) for k in range(2)]
        ctx.set_bases(pts, endomorphism=True)
        r = WholeMsmRunner(m, torch, ctx, n, sets, "endomorphism")
        warm, count = r.group * r.depth + min(W, 3), max(5, r.group * 2)
        el, last, st = timed_run(r, warm, count)
        ctx.set_stage_timing(0)
        whole = ctx.msm(sets[0])
        # the 8 window ranges an 8-GPU run would take (16 full-length windows over the n plain records: no endomorphism split on this path)
        parts = [ctx.msm_windows(sets[0], *window_range(rk, 8)) for rk in range(8)]
        shares_ok = bool(m.MsmContext.combine_windows(torch.cat(parts, dim=0)) == whole)
        k = min(1 << 16, max(n // 4, 1))
        off = (5 << 20) if n > (6 << 20) else n // 2
        sl_p, sl_s = pts[off:off + k].contiguous(), sets[0][off:off + k].contiguous()
        ctx.set_bases(sl_p, endomorphism=True)
        got = ctx.msm(sl_s)
        cfg["c4"] = {"workload": "2^%d BN254 G1 MSM, one GPU, bases with their endomorphism images" % logn, "value": count / el, "unit": "MSM/s",
                     "ms_per_msm": el * 1e3 / count, "timed_msms": count, "untimed_msms_directly_before": warm, "window_bits": st[0][2],
                     "roofline": r.roofline(st), "whole_equals_8_window_range_shares_combined": shares_ok}
        checks.append({"config": "c4", "key": "slice_2p%d_verified_bit_exact_vs_cpu" % (k.bit_length() - 1), "points": host(sl_p), "scalars": host(sl_s),
                       "got": got.to_affine_bytes()})
        del pts, sets, r, parts, sl_p, sl_s

    if "c5" in spec:
        b_, l_ = spec["c5"].split("x")
        batch, n = int(b_), 1 << int(l_)
        pts = ctx.sample_points(n, 0xC50001)
        sc = ctx.sample_scalars(n * batch, 0xC50002)  # `batch` independent scalar vectors, contiguous
        c5 = {"workload": "%d x 2^%s BN254 G1 MSMs over one shared base, one GPU, msm_hip_run_batch_device" % (batch, l_),
              "note": "at 8 GPUs whole MSMs are dealt out (no exchange on the data path): 8 x this figure is the projection, the driver's SCALE run the measurement"}
        results = {}
        for mode in ("endomorphism", "tables_wide"):
            ctx.set_bases(pts, endomorphism=mode == "endomorphism", precompute="wide" if mode == "tables_wide" else False)
            ctx.set_stage_timing(1)
            ctx.msm_batch(sc, n)  # warm-up: one whole batch (the pools take this mode's shape)
            passes = 2
            sync()
            t0 = time.perf_counter()
            for _ in range(passes):
                res = ctx.msm_batch(sc, n)
            sync()
            el = time.perf_counter() - t0
            g = ctx.batch_group_size(n)
            last_g = batch % g or g
            bits = ctx.last_window_bits()
            shp = smvp_shape(m, ctx, n, mode, bits)
            kms = ctx.stage_ms()["smvp"]
            alg = smvp_algorithmic_bytes(shp[0], last_g * shp[1], 1 << (bits - 1))
            results[mode] = [x.to_affine_bytes() for x in res]
            c5[mode] = {"value": passes * batch / el, "unit": "MSM/s", "ms_per_msm": el * 1e3 / (passes * batch), "timed_msms": passes * batch,
                        "untimed_msms_directly_before": batch, "msms_per_launch": g, "window_bits": bits if mode == "endomorphism" else ctx.wide_bits(),
                        "roofline": {"bound": "hbm", "kernel": "k_smvp_chunks", "kernel_ms": kms, "kernel_ms_is": "the batch's last launch (%d MSMs)" % last_g,
                                     "algorithmic_bytes": alg, "achieved": alg / (kms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                     "frac": alg / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS}}
        c5["both_modes_same_results"] = results["endomorphism"] == results["tables_wide"]
        c5["all_results_differ"] = len(set(results["endomorphism"])) == batch
        cfg["c5"] = c5
        kv = batch - 1
        checks.append({"config": "c5", "key": "vector_%d_verified_bit_exact_vs_cpu" % kv, "points": host(pts), "scalars": host(sc[kv * n:(kv + 1) * n]),
                       "got": results["endomorphism"][kv]})
        del pts, sc
    ctx.set_stage_timing(2)
    torch.cuda.empty_cache()
    return cfg, checks


def self_launch(n_ranks):
    """`python bench.py --gpus N` without a launcher: this parent (which never touches the GPU and never execs) starts one
    child per GPU with the torch.distributed environment set, lets rank 0 print the JSON line on the shared stdout, and
    exits with the worst child status.  A failing rank takes the others down (killed by PID) instead of leaving them in a
    collective."""
    import signal
    import socket
    import subprocess

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    deadline = time.monotonic() + float(os.environ.get("BENCH_LAUNCH_TIMEOUT_S", "1500"))  # a rank stuck in a collective ends the run

    def on_signal(signum, frame):  # SIGTERM / SIGINT: the GPU-holding children go with the parent
        raise KeyboardInterrupt

    old = {sig: signal.signal(sig, on_signal) for sig in (signal.SIGTERM, signal.SIGINT)}
    worst = 0
    try:
        for r in range(n_ranks):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
        pending = set(range(n_ranks))
        while pending:
            for r in sorted(pending):
                rc = procs[r].poll()
                if rc is None:
                    continue
                pending.discard(r)
                if rc != 0:
                    worst = worst or rc
                    for q in pending:
                        procs[q].kill()
            if pending and time.monotonic() > deadline:
                sys.stderr.write("bench.py: ranks %s still running at the deadline, killing them\n" % sorted(pending))
                worst = worst or 124
                break
            time.sleep(0.05)
    except KeyboardInterrupt:
        worst = worst or 130
    finally:
        for pr in procs:  # whatever happened: no child outlives the launcher (killed by PID, then reaped)
            if pr.poll() is None:
                pr.kill()
        for pr in procs:
            try:
                pr.wait(timeout=30)
            except Exception:
                pass
        for sig, h in old.items():
            signal.signal(sig, h)
    return worst


def run_native_child(args, bases_mode, n_ranks):
    """rank 0, after its own timed region (every rank idle in a host-side wait): ONE fresh child process times the same window-sharded
    workload through the in-process C ABI a Rust caller gets (native_mgpu_main: msm_hip_mgpu_launch_batch_device /
    finish_batch over `n_ranks` devices, ncclAllGather per launch) -- reported beside the torch.distributed figure as value_native_mgpu.
    A failing child is a non-zero exit of the child and null fields here; nothing of it runs inside this process."""
    import subprocess

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                                                             "GROUP_RANK", "ROLE_RANK", "ROLE_WORLD_SIZE", "GROUP_WORLD_SIZE", "TORCHELASTIC_RUN_ID")}
    env.update(BENCH_MGPU_NATIVE="1", BENCH_BASES=bases_mode)
    if os.environ.get("BENCH_ALL_ON_GPU0") == "1":  # one-GPU rehearsal: the contexts share GPU 0 (pinned-buffer gather)
        env["BENCH_MGPU_IDS"] = ",".join(["0"] * n_ranks)
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", str(n_ranks), "--steps", str(args.steps), "--warmup", str(args.warmup),
           "--logn", str(args.logn), "--no-cpu-baseline"]
    try:
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True,
                           timeout=float(os.environ.get("BENCH_NATIVE_TIMEOUT_S", "600")))
    except subprocess.TimeoutExpired:
        return None, "native child timed out"
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if r.returncode != 0 or len(lines) != 1:
        return None, "native child exit %d: %s" % (r.returncode, r.stderr.strip()[-300:])
    return json.loads(lines[0]), None


def native_mgpu_main(args):
    """BENCH_MGPU_NATIVE=1: the same window-sharded workload through the in-process multi-GPU C ABI (msm_hip_mgpu_launch_batch_device /
    msm_hip_mgpu_finish_batch: one host process, one engine context and one persistent host thread per GPU, one ncclAllGather per
    launch, host combines on the library's pool) instead of one process per GPU over torch.distributed -- so that the path a Rust caller
    gets can be timed next to the one the driver measures.  BENCH_MGPU_IDS=0,0,0,0,0,0,0,0 rehearses it with several contexts on one GPU
    (pinned-buffer gather; the contexts then SHARE that GPU, so the figure is a one-GPU total, not a scaling result)."""
    import torch

    import msm_webgpu_amd as m

    ids = [int(x) for x in os.environ.get("BENCH_MGPU_IDS", ",".join(str(d) for d in range(args.gpus))).split(",")]
    distinct = len(set(ids)) == len(ids)
    n = 1 << args.logn
    bases_mode = os.environ.get("BENCH_BASES") or "plain"
    ctx0 = m.MsmContext(ids[0])
    points = ctx0.sample_points(n, 0x6D736D5F0000 + args.logn)
    mg = m.MultiGpuMsm(ids, "auto" if distinct else "host")
    if bases_mode == "tables_wide":
        mg.set_wide_bits(int(os.environ.get("BENCH_WIDE_BITS", "19")))
    mg.set_bases(points.cpu().numpy().tobytes(), endomorphism=bases_mode == "endomorphism", precompute="wide" if bases_mode == "tables_wide" else False)
    full = mg.group_size
    group = int(os.environ.get("BENCH_MSMS_PER_LAUNCH", "0")) or full
    per_dev = {}
    for d in set(ids):  # identical synthetic scalars resident on every device (the deterministic device sampler)
        c = ctx0 if d == ids[0] else m.MsmContext(d)
        sets = [c.sample_scalars(n, 0x6D736D5F1000 + args.logn + 7 * i) for i in range(2)]
        per_dev[d] = torch.cat([sets[k & 1] for k in range(group)], dim=0).contiguous()
        if c is not ctx0:
            c.close()
    scal = [per_dev[d] for d in ids]
    depth = max(1, min(3, int(os.environ.get("BENCH_PIPE_DEPTH", "3"))))

    def run_steps(count):
        k = -(-count // group)
        sizes = [count // k + (1 if i < count % k else 0) for i in range(k)]
        pending, last = [], None
        for i, gs in enumerate(sizes):
            slot = i % (depth + 1)
            mg.launch_batch([t[: gs * n] for t in scal], n, slot, inputs_complete=True)  # sampled and synchronised before the timed region
            pending.append((slot, gs))
            if len(pending) == depth:
                s0, g0 = pending.pop(0)
                last = mg.finish_batch(s0, g0)
        for s0, g0 in pending:
            last = mg.finish_batch(s0, g0)
        return last

    run_steps(int(os.environ.get("BENCH_STEADY_MSMS", "40")))
    run_steps(max(args.warmup, 1))
    for d in set(ids):
        torch.cuda.synchronize(d)
    t0 = time.perf_counter()
    last = run_steps(args.steps)
    for d in set(ids):
        torch.cuda.synchronize(d)
    elapsed = time.perf_counter() - t0
    ctx0.set_bases(points, endomorphism=bases_mode == "endomorphism")
    sets0 = [ctx0.sample_scalars(n, 0x6D736D5F1000 + args.logn + 7 * i) for i in range(2)]
    ok = bool(ctx0.msm(sets0[(len(last) - 1) & 1]) == last[-1])
    print(json.dumps({"metric": "BN254 MSM/s at 2^%d points" % args.logn, "value": args.steps / elapsed, "unit": "MSM/s", "n_gpus": len(ids),
                      "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed * 1e3 / args.steps, "higher_is_better": True,
                      "scaling": "strong", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
                      "config": {"workload": "2^%d BN254 G1 MSM, 16-bit signed-bucket windows, inputs resident in HBM" % args.logn,
                                 "parallelism": "in-process msm_hip_mgpu_*: %s windows over %d contexts, %s gather" % (
                                     "8 half-length" if bases_mode == "endomorphism" else "the virtual (wide tables)" if bases_mode == "tables_wide" else "16", len(ids),
                                     "RCCL" if mg.uses_rccl else "pinned-buffer"),
                                 "device_ids": ids, "msms_per_launch": group, "launches_in_flight": depth},
                      "sharded_result_equals_single_gpu": ok, "native_mgpu": True, "rccl_ranks": len(ids) if mg.uses_rccl else 0,
                      "pre_timed_msms": int(os.environ.get("BENCH_STEADY_MSMS", "40")) + max(args.warmup, 1)}))
    mg.close()
    ctx0.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--logn", type=int, default=20, help="log2 of the MSM size (20 = the config the metric is quoted on)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-logn", type=int, default=None, help="bounded CPU sample size (default: min(logn, 20))")
    args = ap.parse_args()

    if os.environ.get("BENCH_MGPU_NATIVE") == "1" and "RANK" not in os.environ:
        return native_mgpu_main(args)
    if args.gpus > 1 and "RANK" not in os.environ:  # no launcher around us: be the launcher
        raise SystemExit(self_launch(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))

    global torch, dist
    import torch
    import torch.distributed as dist

    # BENCH_DRY_RUN=1: rendezvous rehearsal without a GPU (the CPU test of the launcher): every rank joins a gloo group,
    # contributes its id to one all-gather, rank 0 prints what it saw.  Measures nothing.
    if os.environ.get("BENCH_DRY_RUN"):
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        ids = [torch.zeros(1, dtype=torch.int32) for _ in range(world)]
        dist.all_gather(ids, torch.tensor([rank], dtype=torch.int32))
        if os.environ["BENCH_DRY_RUN"] == "fail_last" and rank == world - 1:
            raise SystemExit(3)
        if os.environ["BENCH_DRY_RUN"] == "hang_last" and rank == world - 1:
            time.sleep(600)  # a rank that never returns: the self-launcher's deadline must end the run
        dist.barrier()
        if rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": world, "dist_backend": "gloo", "dist_ranks": sorted(int(t.item()) for t in ids)}))
        dist.destroy_process_group()
        return

    import msm_webgpu_amd as m  # fails loudly if libmsm_hip.so is missing
    from msm_webgpu_amd.sharding import ShardedMsmPipeline, msms_per_launch, window_range

    assert torch.cuda.is_available(), "bench.py needs a GPU"
    # rehearsal aid for a one-GPU box: BENCH_ALL_ON_GPU0=1 puts every rank on GPU 0 (with BENCH_DIST_BACKEND=gloo, since
    # RCCL refuses two ranks on one device) -- checks the multi-rank logic end to end; its timings mean nothing
    if os.environ.get("BENCH_ALL_ON_GPU0") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    # BENCH_FORCE_SHARDED=1 drives the multi-GPU code path (RCCL process group + sharded pipeline) even with one rank,
    # so that it can be exercised on a single-GPU box under torch.distributed.run --nproc-per-node 1
    force_sharded = os.environ.get("BENCH_FORCE_SHARDED") == "1"
    use_dist = world > 1 or (force_sharded and "RANK" in os.environ)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    # host-side waits beside RCCL's collectives: where ranks must WAIT for one another without occupying their GPUs -- an RCCL barrier is a
    # kernel that spins on every GPU until the last rank arrives (rank 0's native child, below, uses all of them meanwhile) -- they meet at
    # the rendezvous store (counters and keys; no second process group, no second transport)
    def host_wait(tag):
        import datetime

        store = dist.distributed_c10d._get_default_store()
        store.add("bench_%s_arrived" % tag, 1)
        if rank == 0:
            while store.add("bench_%s_arrived" % tag, 0) < world:
                time.sleep(0.005)
            store.set("bench_%s_go" % tag, "1")
        else:
            store.wait(["bench_%s_go" % tag], datetime.timedelta(minutes=30))

    # ranks that really take part in the collectives: every rank contributes its id to one all-gather
    dist_ranks, dist_backend = None, None
    if use_dist:
        dist_backend = dist.get_backend()
        ids = torch.full((world,), -1, dtype=torch.int32, device="cuda")
        dist.all_gather_into_tensor(ids, torch.tensor([rank], dtype=torch.int32, device="cuda"))
        dist_ranks = int((ids.cpu() == torch.arange(world, dtype=torch.int32)).sum().item())
        assert dist_ranks == dist.get_world_size() == world

    n = 1 << args.logn
    ctx = m.MsmContext(local_rank)
    # identical synthetic inputs on every rank (deterministic device sampler), resident in HBM
    points = ctx.sample_points(n, 0x6D736D5F0000 + args.logn)
    scalar_sets = [ctx.sample_scalars(n, 0x6D736D5F1000 + args.logn + 7 * i) for i in range(2)]
    # tuning aid (never a reported result): BENCH_EMULATE_WORLD=8 makes this single rank do the per-rank share of an
    # 8-rank run (2 windows) through the sharded pipeline; the MSM value is then NOT a whole-job figure
    emulate = int(os.environ.get("BENCH_EMULATE_WORLD", "0"))
    sharded = world > 1 or force_sharded or emulate > 1
    # how the resident bases are held (include/msm_hip.h; the result is the same group element in every mode):
    #   endomorphism  P_i and phi(P_i): every scalar is split into two 127-bit halves on the device, 8 windows over 2n points
    #                 (default on one GPU: the same bucket additions, half the buckets to reduce; 2 x the base memory)
    #   plain         the reference's shape: 16 windows over n points
    #   tables        fixed-base tables 2^(16 w) P_i: one bucket set per MSM (16 x the base memory)
    #   tables_wide   fixed-base tables 2^(C w) P_i, C-bit digits (17 up to 2^20 points: 15 bucket additions per point into 2 virtual windows of 2^15
    #                 slots; 20 beyond: 13 additions, 16 virtual windows).  Window-sharded runs: the ranks share the VIRTUAL windows
    #                 (BENCH_WIDE_BITS, default 19: 8 of them, 14 additions per point)
    #   (window-sharded runs: plain by default -- with endomorphism bases the ranks share the 8 half-length windows, one per rank at
    #    8 GPUs, measured 8 % slower per MSM than two full-length windows per rank: every rank splits every scalar, profiles/r03_share_ab.txt)
    bases_mode = os.environ.get("BENCH_BASES") or ("plain" if sharded else "endomorphism")
    assert bases_mode in ("plain", "endomorphism", "tables", "tables_wide") and (bases_mode != "tables" or not sharded)
    wide_shares = sharded and bases_mode == "tables_wide"
    if wide_shares:
        ctx.set_wide_bits(int(os.environ.get("BENCH_WIDE_BITS", "19")))
    ctx.set_bases(points, endomorphism=bases_mode == "endomorphism", precompute="wide" if bases_mode == "tables_wide" else bases_mode == "tables")
    halves = bases_mode == "endomorphism"
    # the windows the ranks share: 16, the 8 half-length ones, or the virtual windows of the wide tables
    shard_windows = (1 << (ctx.wide_bits() - 16)) if wide_shares else NUM_WINDOWS // 2 if halves else NUM_WINDOWS
    w_begin, w_end = window_range(rank, world, shard_windows)
    w_local = w_end - w_begin

    def sync_all():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # window-sharded runs put the shares of several independent MSMs through one launch (as many as make up one MSM's worth of
    # bucket sets: 8 MSMs x 2 windows -- or x 1 half-length window -- at 8 GPUs): one kernel sequence and one RCCL all-gather per
    # group.  (Smaller groups for a short run -- 5 launches of 4 instead of 3 of 7 at the driver's --steps 20 -- were measured and lose:
    # 0.235 vs 0.221 ms per MSM, profiles/r03_share_ab.txt; a launch's fixed costs outweigh the shorter exposed tail.)
    # BENCH_MSMS_PER_LAUNCH overrides.
    group, pipe, runner, combine_mode = 1, None, None, None
    if sharded:
        full = msms_per_launch(emulate if emulate > 1 else world, shard_windows)
        group = int(os.environ.get("BENCH_MSMS_PER_LAUNCH", "0")) or full
        # every MSM's host window combine runs ONCE across the ranks (vector v of a launch on rank v % world), not once per rank
        combine_mode = os.environ.get("BENCH_COMBINE", "spread")
        pipe = ShardedMsmPipeline(ctx, rank, world, depth=int(os.environ.get("BENCH_PIPE_DEPTH", "3")), msms_per_issue=group,
                                  emulate_world=emulate, halves=halves, combine=combine_mode, wide=wide_shares)
        w_local = pipe.w_end - pipe.w_begin
        # vector k of a group is scalar set k & 1
        group_scalars = torch.cat([scalar_sets[k & 1] for k in range(group)], dim=0).contiguous() if group > 1 else None
    else:
        runner = WholeMsmRunner(m, torch, ctx, n, scalar_sets, bases_mode)

    def sharded_sizes(count):
        k = -(-count // group)
        return [count // k + (1 if i < count % k else 0) for i in range(k)]

    def run_steps(count, stages=None):
        """`count` MSMs; returns the last result.  N = 1 pipelines the host combine of MSM i with the device work of i+1.
        `stages` receives (smvp kernel ms, bucket sets, window bits) of every launch."""
        if not sharded:
            return runner.run(count, stages)
        # windows sharded over the ranks; device work, RCCL all-gather, D2H and host combine all pipelined
        result, inflight = None, []

        def collect():
            res = pipe.complete()
            if stages is not None:
                stages.append((ctx.stage_ms()["smvp"], inflight[0] * w_local, 16))
            inflight.pop(0)
            return res

        for k, gs in enumerate(sharded_sizes(count)):
            if group > 1:
                pipe.issue(group_scalars[: gs * n], n, inputs_complete=True)  # sampled and synchronised before the timed region
            else:
                pipe.issue(scalar_sets[k & 1], inputs_complete=True)
            inflight.append(gs)
            if len(inflight) == pipe.depth:
                result = collect()
        while inflight:
            result = collect()
        return result

    def timed(count, stages):
        """barrier + synchronise, `count` steps, barrier + synchronise; the MAX over the ranks"""
        sync_all()
        t0 = time.perf_counter()
        res = run_steps(count, stages)
        sync_all()
        el = time.perf_counter() - t0
        if use_dist:
            t = torch.tensor([el], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el, res

    # ---- THE MEASUREMENT, under the literal protocol: W warm-up steps on a GPU that has done nothing but sample the inputs and convert the
    # bases, then EXACTLY K timed steps -> `value`.  HIP events only around the SMVP accumulate kernel (the roofline figure; every extra
    # stage event costs queue time between kernels).  (Rounds 1 - 4 reported as `value` a second timed region that followed ~45 further
    # untimed steps: after an idle phase the GPU needs ~15 MSMs, ~20 ms of this work, to reach its steady state -- tools/step_time_trend.py:
    # first 20 MSMs 1.53 ms each after 5 warm-up steps, 1.36 from the 21st on.  That figure is still measured below and reported as
    # `value_steady_state`, with the number of untimed steps in front of it.)
    ctx.set_stage_timing(1)
    stages = []
    run_steps(max(args.warmup, 1), None)
    elapsed, last = timed(args.steps, stages)
    pre_timed_msms = max(args.warmup, 1)  # MSMs (or MSM shares) run before `value`'s timed region
    ctx.set_stage_timing(2)

    # timing scopes B and C of SURVEY.md section 8(d), informational (never `value`): B = scalars arrive from host memory
    # (32 MiB H2D per MSM at 2^20), bases resident -- as the latency of one call and as the throughput of three slots in rotation
    # (msm_hip_launch: the copy of MSM i+1 runs on the copy stream under the device work of MSM i); C = one-shot incl.
    # base upload (≙ the reference's compute_msm call shape): the first call also creates the context the library then keeps
    scope_ms = None
    # (not under rocprofv3: since round 5 these calls run as sub-MSMs over halves of the points -- half-size k_smvp_chunks launches that would mix into
    #  the per-kernel averages and counter sums the profile of this command is read for)
    if world == 1 and emulate <= 1 and args.logn <= 22 and "ROCP_TOOL_LIBRARIES" not in os.environ:
        sb_host = [s.cpu().numpy().tobytes() for s in scalar_sets]
        pb_host = points.cpu().numpy().tobytes()
        import ctypes

        one = ctypes.create_string_buffer(96)
        tc = []
        for _ in range(4):
            t1 = time.perf_counter()
            rc1 = m.lib().msm_hip_msm_bn254_g1(pb_host, sb_host[0], n, one)  # uploads the bases, runs; the library keeps its context
            tc.append((time.perf_counter() - t1) * 1e3)
            assert rc1 == 0, rc1
        m.lib().msm_hip_oneshot_release()
        tb = []
        for i in range(5):
            t1 = time.perf_counter()
            ctx.msm(sb_host[i & 1])
            tb.append((time.perf_counter() - t1) * 1e3)
        k, slots = 24, 3  # three result slots in flight: the copy of MSM i+2 and the sort of i+1 under the SMVP of i
        ctx.set_stage_timing(0)  # (the stage events of the latency runs above cost queue time between kernels)
        for timed_pass in (False, True):  # the first pass brings the GPU out of its idle state
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for i in range(k):
                if i >= slots:
                    ctx.finish(i % slots)
                ctx.launch_host(sb_host[i & 1], i % slots)
            for i in range(k, k + slots):
                ctx.finish(i % slots)
            b_pipe = (time.perf_counter() - t1) * 1e3 / k
        scope_ms = {"B_host_scalars_resident_bases_latency": sorted(tb)[2], "B_host_scalars_three_slots_pipelined": b_pipe,
                    "C_one_shot_with_base_upload": sorted(tc[1:])[1], "C_one_shot_first_call": tc[0]}
        del sb_host, pb_host

    # window-sharded runs: the latency of ONE MSM across the ranks (its window shares, the gather, the host combine; nothing in
    # flight beside it) -- BASELINE config 3 as a single call -- median of 20
    sharded_latency_ms = None
    if sharded:
        ctx.set_stage_timing(0)
        lat = []
        for i in range(20):
            sync_all()
            t1 = time.perf_counter()
            pipe.issue(scalar_sets[i & 1], inputs_complete=True)
            pipe.complete()
            lat.append((time.perf_counter() - t1) * 1e3)
        sharded_latency_ms = sorted(lat)[len(lat) // 2]

    # ---- the same K steps in the steady state (informational: `value_steady_state`): BENCH_STEADY_MSMS (default 40) steps of the timed
    # workload itself and the W warm-up steps run back to back right before them
    ctx.set_stage_timing(1)
    steady = max(0, int(os.environ.get("BENCH_STEADY_MSMS", "40")))
    steady_stages = []
    if steady:
        run_steps(steady, None)
    run_steps(max(args.warmup, 1), None)
    steady_elapsed, _ = timed(args.steps, steady_stages)

    # single-MSM latency (no pipelining, stages not overlapped) -- reported beside the throughput figure.  Measured WITHOUT stage events (what
    # a caller of the synchronous entry point sees: every HIP event between two kernels costs queue time, ~45 us over the twelve kernels of a
    # 2^16 MSM); a second pass with the events on gives the per-stage breakdown and its own, longer, latency.
    latency_ms, latency_events_ms, isolated = None, None, None
    if world == 1 and emulate <= 1:
        for level in (0, 2):
            ctx.set_stage_timing(level)
            lat = []
            for i in range(9 if level == 0 else 5):
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                ctx.msm(scalar_sets[i & 1])
                lat.append((time.perf_counter() - t1) * 1e3)
                if level:
                    isolated = ctx.stage_ms()
            if level == 0:
                latency_ms = sorted(lat[2:])[len(lat[2:]) // 2]
            else:
                latency_events_ms = sorted(lat)[len(lat) // 2]
    ctx.set_stage_timing(2)

    # sharded runs: check the gathered + combined results of the last launch against this rank's own whole MSM (outside the timed region);
    # a rank checks the vectors it combined ("spread": vector v on rank v % world), the verdict is the AND over the ranks
    sharded_ok = None
    if sharded and emulate <= 1:
        res = last if isinstance(last, list) else [last]  # vector k of a group used scalar set k & 1
        whole, sharded_ok = {}, True
        for k, g1 in enumerate(res):
            if g1 is None:
                continue
            idx = (k & 1) if group > 1 else ((args.steps - 1) & 1)
            if idx not in whole:
                whole[idx] = ctx.msm(scalar_sets[idx])
            sharded_ok = sharded_ok and bool(whole[idx] == g1)
        if use_dist:
            t = torch.tensor([1 if sharded_ok else 0], dtype=torch.int32, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            sharded_ok = bool(t.item())

    # (not under rocprofv3 unless asked for -- BENCH_TABLES_WIDE=2 --: the launches of the blocks below would mix into the per-kernel
    #  averages of the profile this command is compared with)
    profiled = "ROCP_TOOL_LIBRARIES" in os.environ and os.environ.get("BENCH_TABLES_WIDE") != "2"
    single = world == 1 and emulate <= 1 and not profiled

    # The opt-in fixed-base mode beside the headline (informational, never `value`; BENCH_TABLES_WIDE=0 skips it): the same scalars through the
    # same pipeline with the bases held as wide tables (MSM_HIP_BASES_PRECOMPUTE_WIDE: 15 bucket additions per point at 2^20 instead of 16,
    # 13 from 2^22 up), after everything the headline needs has been measured; its result must be the headline mode's.
    wide_line = None
    if (single and bases_mode != "tables_wide" and os.environ.get("BENCH_TABLES_WIDE", "1") != "0"
            and args.logn <= (24 if os.environ.get("BENCH_TABLES_WIDE") == "2" else 22)):  # (2^23, 2^24: 13 GiB of tables + 31 GiB of sort arrays, on request)
        ctx.set_stage_timing(0)
        want = ctx.msm(scalar_sets[0])     # the headline mode's result
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        ctx.set_bases(points, precompute="wide")
        torch.cuda.synchronize()
        setup_ms = (time.perf_counter() - t1) * 1e3
        wrun = WholeMsmRunner(m, torch, ctx, n, scalar_sets, "tables_wide")  # the grouping of small MSMs follows the mode: an MSM is 2 local windows here, not 8 - 10
        same = bool(ctx.msm(scalar_sets[0]) == want)
        ctx.set_stage_timing(1)
        wrun.run(wrun.group * wrun.depth, None)  # one full-size launch through every slot first: the pools grow to this mode's shape outside the timed region
        wrun.run(steady // 2 + max(args.warmup, 1), None)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        wide_stages = []
        wrun.run(args.steps, wide_stages)
        torch.cuda.synchronize()
        wide_elapsed = time.perf_counter() - t1
        wide_line = {"value": args.steps / wide_elapsed, "unit": "MSM/s", "ms_per_step": wide_elapsed * 1e3 / args.steps, "digit_bits": ctx.wide_bits(),
                     "msms_per_launch": wrun.group, "table_setup_ms": setup_ms, "same_result_as_headline_mode": same,
                     "roofline": wrun.roofline(wide_stages),  # this mode's SMVP launches over its own timed steps (fewer entries per point than the headline's)
                     "untimed_steps_before_timed_region": wrun.group * wrun.depth + steady // 2 + max(args.warmup, 1),
                     "note": "opt-in MSM_HIP_BASES_PRECOMPUTE_WIDE (fixed bases: 13 - 15 x the base memory); steady-state protocol (compare with value_steady_state), measured after the headline"}
        ctx.set_stage_timing(2)
        del wrun
        ctx.set_bases(points, endomorphism=bases_mode == "endomorphism", precompute=bases_mode == "tables")  # the headline's bases again

    # ---- BASELINE.json's other configs on the same line (`configs`; informational, never `value`): C1 2^16, C3 one rank's share of 8 at 2^20,
    # C4 2^24, C5 64 x 2^18 over one base -- each with its own timed region, SMVP roofline fraction and an in-run check of its result.
    # Default run only (--logn 20, one GPU); BENCH_CONFIGS=0 skips it, BENCH_CONFIGS="c1:10,c3:12,c4:13,c5:6x11" picks other sizes (tests).
    configs, config_checks = None, []
    cfg_spec = os.environ.get("BENCH_CONFIGS", "c1:16,c3:20,c4:24,c5:64x18" if args.logn == 20 else "0")
    if single and cfg_spec != "0":
        configs, config_checks = measure_configs(m, torch, ctx, args, cfg_spec, points if "c3:%d" % args.logn in cfg_spec else None, scalar_sets)
        ctx.set_bases(points, endomorphism=bases_mode == "endomorphism", precompute="wide" if bases_mode == "tables_wide" else bases_mode == "tables")

    ms_per_step = elapsed * 1e3 / args.steps
    # roofline of the SMVP accumulate kernel: algorithmic bytes of all timed launches / their summed durations
    # (geometry of every timed launch from the window size the engine reports for it: grouped small MSMs run 14-bit windows)
    def roofline_of(st):
        launches = len(st)
        avg_ms = sum(s[0] for s in st) / launches
        if sharded:
            n_in = 2 * n if halves else n  # inputs per bucket set; wide shares: the 14 n digit entries spread over the virtual windows
            per_set = (lambda b: ((254 + ctx.wide_bits()) // ctx.wide_bits()) * n / shard_windows) if wide_shares else (lambda b: n_in)
        else:
            per_set = lambda b: smvp_shape(m, ctx, n, bases_mode, b)[0]
        alg = sum(smvp_algorithmic_bytes(per_set(b), w, 1 << (b - 1)) for _, w, b in st) / launches
        ach = alg / (avg_ms * 1e-3) / 1e9
        mads = sum(per_set(b) * w for _, w, b in st) * MADS_PER_MIXED_ADD / (sum(s[0] for s in st) * 1e-3)
        return avg_ms, alg, ach, mads

    smvp_avg_ms, alg_bytes, achieved, lane_mads_per_s = roofline_of(stages)
    st_ms, st_alg, st_ach, _ = roofline_of(steady_stages)
    w_launch = max(s[1] for s in stages)
    bits_all = [s[2] for s in stages]
    bits_main = max(set(bits_all), key=bits_all.count)
    # the rocprofv3 --kernel-trace --stats average of the same kernel in the same command, when this round's profile is committed
    kernel_ms_rocprof, rocprof_source = None, None
    try:
        with open(os.path.join(ROOT, "profiles", "rocprof_kernel_ms.json")) as f:
            rp = json.load(f)
        key = "logn%d_%s_%s" % (args.logn, bases_mode, "w%d" % w_local if sharded else "single")
        if key in rp:
            kernel_ms_rocprof, rocprof_source = rp[key]["k_smvp_chunks_avg_ms"], rp[key]["source"] + " (a committed profile of the same command, not measured in this run)"
    except (OSError, ValueError, KeyError):
        pass

    traffic, traffic_source = None, None
    try:
        pmc_name = "smvp_pmc_traffic.json" if args.logn == 20 else "smvp_pmc_traffic_logn%d.json" % args.logn
        with open(os.path.join(ROOT, "profiles", pmc_name)) as f:
            pmc = json.load(f)
        if pmc.get("logn") == args.logn and pmc.get("w_local") == w_launch and pmc.get("bases", "plain") == bases_mode:  # one whole-MSM launch
            traffic = pmc.get("hbm_bytes_per_launch")
            traffic_source = "profiles/%s: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this workload, calibrated (profiles/README.md); not measured in this run" % pmc_name
    except (OSError, ValueError):
        pass

    shape1 = None if sharded else smvp_shape(m, ctx, n, bases_mode, bits_main)
    out = {
        "metric": "BN254 MSM/s at 2^%d points" % args.logn,
        "value": args.steps / elapsed,
        "unit": "MSM/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "u32",
        "data": "synthetic",
        "protocol": "literal: inputs sampled and bases converted, W warm-up steps, barrier + synchronise, K timed steps, barrier + synchronise",
        "untimed_steps_before_timed_region": max(args.warmup, 1),
        "value_steady_state": args.steps / steady_elapsed,
        "ms_per_step_steady_state": steady_elapsed * 1e3 / args.steps,
        "steady_state": {"untimed_steps_of_this_workload_directly_before": steady + max(args.warmup, 1), "smvp_kernel_ms": st_ms, "roofline_frac": st_ach / HBM_PEAK_GBS,
                         "note": "the same K steps after BENCH_STEADY_MSMS further steps (and the scope / latency blocks): what rounds 1 - 4 reported as value"},
        "fixed_base_tables_wide": wide_line,
        "configs": configs,
        "config": {"workload": "2^%d BN254 G1 MSM, 16-bit signed-bucket windows, inputs resident in HBM" % args.logn,
                   "bases": {"plain": "n points, 16 windows (the reference's shape)",
                             "endomorphism": "P and phi(P) resident: scalars split into two 127-bit halves on the device, 8 windows over 2n points",
                             "tables": "fixed-base tables 2^(16 w) P resident: one bucket set per MSM",
                             "tables_wide": "fixed-base tables 2^(C w) P resident: ceil(255 / C) digits of C bits per scalar (C = 16 up to 2^16 points, 17 up to 2^20, 20 beyond), "
                                            "one bucket set of 2^(C-1) slots run as virtual windows of 2^15"}[bases_mode],
                   "window_bits": bits_main,
                   "windows_per_gpu": w_local if sharded else shape1[1], "msms_per_launch": group if sharded else runner.group,
                   "parallelism": ("%s windows/%d + RCCL all-gather" % ("%d virtual (wide tables, %d-bit digits)" % (shard_windows, ctx.wide_bits()) if wide_shares
                                                                       else "8 half-length" if halves else "16", world)) if sharded else "single GPU",
                   "launches_in_flight": pipe.depth if sharded else runner.depth,
                   "host_combine": "pipelined behind the device work of the following launches",
                   "combine": ({"mode": combine_mode, "owner": {"spread": "vector v of a launch is combined once, on rank v % world (the other ranks return None for it)",
                                                                "all": "every rank combines every MSM", "rank0": "rank 0 combines everything"}[combine_mode]}
                               if sharded else None)},
        "roofline": {"bound": "hbm", "kernel": "k_smvp_chunks", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source, "algorithmic_bytes": alg_bytes,
                     "kernel_ms": smvp_avg_ms, "kernel_ms_rocprof": kernel_ms_rocprof, "kernel_ms_rocprof_source": rocprof_source},
        # beside (not instead of) the HBM figure: the kernel's multiply-add rate against the instruction's measured issue peak
        "roofline_valu": {"bound": "valu_issue", "kernel": "k_smvp_chunks", "unit": "T lane-mad/s (v_mad_u64_u32)",
                          "achieved": lane_mads_per_s / 1e12, "peak": VALU_PEAK_LANE_MADS / 1e12, "frac": lane_mads_per_s / VALU_PEAK_LANE_MADS},
        "smvp_ms_pipelined": smvp_avg_ms,
        "rccl_ranks": dist_ranks if dist_backend == "nccl" else None,
        "dist_backend": dist_backend,
        "dist_ranks": dist_ranks,
        "emulated_world": emulate if emulate > 1 else None,
        "sharded_result_equals_single_gpu": sharded_ok,
        "latency_ms_single_msm": latency_ms if not sharded else sharded_latency_ms,
        "latency_ms_single_msm_with_stage_events": latency_events_ms,
        "stage_ms_single_msm": isolated,
        "scope_ms": scope_ms,
        "pre_timed_msms": pre_timed_msms,
    }

    # N > 1: the same workload through the in-process multi-GPU C ABI (what a Rust caller of src/lib.rs:76-82 links against), timed by ONE
    # fresh child of rank 0 while every rank of this job idles in a host-side wait -- value stays the torch.distributed figure
    if use_dist and world > 1 and emulate <= 1 and os.environ.get("BENCH_NATIVE_CHILD", "1") != "0":
        host_wait("timed_region_left")  # every rank has left its timed region: the GPUs are idle
        child, err = None, None
        if rank == 0:
            child, err = run_native_child(args, bases_mode, world)
            out["value_native_mgpu"] = child["value"] if child else None
            out["ms_per_step_native_mgpu"] = child["ms_per_step"] if child else None
            out["native_rccl_ranks"] = child.get("rccl_ranks") if child else None
            out["native_mgpu"] = ({"config": child["config"], "result_equals_single_gpu": child["sharded_result_equals_single_gpu"],
                                   "pre_timed_msms": child.get("pre_timed_msms")} if child else {"error": err})
        host_wait("native_child_done")

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import cpu as oracle_cpu  # the checker + the timed CPU baseline; never on the product path

        logs = args.cpu_sample_logn if args.cpu_sample_logn is not None else min(args.logn, 20)
        ns = 1 << logs
        pb = points[:ns].cpu().numpy().tobytes()
        last_set = scalar_sets[(args.steps - 1) & 1 if sharded else runner.last_set(args.steps)]
        sb = last_set[:ns].cpu().numpy().tobytes()
        t1 = time.perf_counter()
        want = oracle_cpu.cpu_msm(pb, sb, 1)
        cpu_s = time.perf_counter() - t1
        out["cpu_baseline"] = {"value": (ns / n) / cpu_s, "unit": "MSM/s", "cores": 1, "kind": "port",
                               "sample": "one 2^%d MSM on 1 thread (oracle/bn254.c restatement of halo2curves msm_serial; "
                                         "scaled by 2^%d/2^%d)" % (logs, logs, args.logn),
                               "seconds": cpu_s}
        if ns == n:
            last_g1 = last[-1] if isinstance(last, list) else last
            out["verified_bit_exact_vs_cpu"] = bool(last_g1.to_affine_bytes() == oracle_cpu.to_affine64(want))
        threads = min(os.cpu_count() or 1, 64)
        t1 = time.perf_counter()
        want_mt = oracle_cpu.cpu_msm(pb, sb, threads)
        cpu_mt = time.perf_counter() - t1
        out["cpu_baseline_mt"] = {"value": (ns / n) / cpu_mt, "unit": "MSM/s", "cores": threads, "kind": "port", "seconds": cpu_mt,
                                  "agrees": bool(oracle_cpu.to_affine64(want_mt) == oracle_cpu.to_affine64(want))}
        # the configs' results against the same oracle (and, for C1 -- BASELINE's CPU-path config --, the oracle's own time)
        for chk in config_checks:
            t1 = time.perf_counter()
            w = oracle_cpu.to_affine64(oracle_cpu.cpu_msm(chk["points"], chk["scalars"], chk.get("threads", threads)))
            secs = time.perf_counter() - t1
            configs[chk["config"]][chk["key"]] = bool(w == chk["got"])
            if chk.get("time_as"):
                configs[chk["config"]][chk["time_as"]] = {"value": 1.0 / secs, "unit": "MSM/s", "cores": chk.get("threads", threads), "kind": "port", "seconds": secs}
    elif configs:
        for chk in config_checks:
            configs[chk["config"]][chk["key"]] = None  # --no-cpu-baseline: not checked

    if rank == 0:
        print(json.dumps(out))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()



if __name__ == "__main__":
    main()
