"""The same engine on every curve it is built for: pipelined MSM time at 2^logn (endomorphism bases, two launches in flight, as
bench.py's single-GPU line), single-MSM latency, SMVP kernel time; one result per curve checked bit-exactly against that curve's oracle.
usage: curve_throughput.py [logn [curve ...]]   (test infrastructure: uses the oracle)"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import msm_webgpu_amd as m

logn = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << logn
for curve in (sys.argv[2:] or ["bn254", "grumpkin", "pallas", "vesta", "bls12_381", "bn254_g2", "bls12_381_g2"]):
    ctx = m.MsmContext(0, curve=curve)
    sc = [ctx.sample_scalars(n, 2 + i) for i in range(2)]
    if curve.endswith("_g2"):
        # G2 (coordinates in Fq2: csrc/fq2.h): the device sampler draws P_i = (a + i b) G -- points of G2 proper -- and the expected result is the
        # closed form (sum_i s_i m_i mod r) G (oracle/bn254_g2_ref.py: sample_multipliers).  CURVE_BASES = endomorphism (default) | plain | tables | tables_wide (every curve)
        g2 = importlib.import_module("oracle." + curve + "_ref")
        pts = ctx.sample_points(n, 1)
        mode = os.environ.get("CURVE_BASES", "endomorphism")
        ctx.set_bases(pts, precompute="wide" if mode == "tables_wide" else mode == "tables", endomorphism=mode == "endomorphism")
    else:
        cpu = importlib.import_module("oracle.cpu" if curve == "bn254" else "oracle.cpu_" + curve)
        pts = ctx.sample_points(n, 1)
        # (BLS12-381's cofactor is not 1: the samplers' curve points are outside the order-r subgroup, where the endomorphism mode is not exact)
        mode = os.environ.get("CURVE_BASES", "endomorphism")
        if mode in ("tables", "tables_wide"):
            ctx.set_bases(pts, precompute="wide" if mode == "tables_wide" else True)
        else:
            ctx.set_bases(pts, endomorphism=curve != "bls12_381" and mode == "endomorphism")
    ctx.set_stage_timing(1)
    def run(k):
        fl, last = [], None
        for j in range(k):
            if len(fl) == 2:
                last = ctx.finish(fl.pop(0))
            ctx.launch(sc[j & 1], j % 2)
            fl.append(j % 2)
        for s in fl:
            last = ctx.finish(s)
        return last
    run(30)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    last = run(100)
    torch.cuda.synchronize()
    step = (time.perf_counter() - t0) * 10
    smvp = ctx.stage_ms()["smvp"]
    lat = []
    for i in range(5):
        torch.cuda.synchronize(); t1 = time.perf_counter(); ctx.msm(sc[i & 1]); lat.append((time.perf_counter() - t1) * 1e3)
    if curve.endswith("_g2"):
        want = g2.affine_to_bytes(g2.msm_by_multipliers(g2.sample_multipliers(n, 1), g2.bytes_to_scalars(sc[1].cpu().numpy().tobytes())))
    else:
        want = cpu.to_affine64(cpu.cpu_msm(pts.cpu().numpy().tobytes(), sc[1].cpu().numpy().tobytes(), min(os.cpu_count() or 1, 32)))
    print("%-12s %s 2^%d: %.4f ms per MSM pipelined (%.0f MSM/s), SMVP kernel %.3f ms, latency %.3f ms, bit-exact vs its oracle: %s"
          % (curve, os.environ.get("CURVE_BASES", "endomorphism") + (" (%d-bit digits)" % ctx.wide_bits() if ctx.wide_bits() else ""), logn, step, 1e3 / step, smvp, sorted(lat)[2], last.to_affine_bytes() == want), flush=True)
    ctx.close()
