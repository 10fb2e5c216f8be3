"""How the engine behaves on heavily skewed scalars (all equal -> one bucket per window holds every entry)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import msm_webgpu_amd as m
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 20
mode = sys.argv[2] if len(sys.argv) > 2 else "plain"   # endomorphism: bases with their images (8 windows over 2n points); tables / tables_wide: fixed-base tables
endo = mode == "endomorphism"
n = 1 << logn
ctx = m.MsmContext(0)
pts = ctx.sample_points(n, 1)
ctx.set_bases(pts, endomorphism=endo, precompute="wide" if mode == "tables_wide" else mode == "tables")
print("bases:", mode)
uni = ctx.sample_scalars(n, 2)
ctx.msm(uni)
s = 0x123456789ABCDEF013579BDF2468ACE0FEDCBA9876543210
eq = torch.tensor(list(s.to_bytes(32, "little")), dtype=torch.uint8, device="cuda").repeat(n, 1).contiguous()
half = uni.clone(); half[: n // 2] = eq[: n // 2]
small = torch.zeros((n, 32), dtype=torch.uint8, device="cuda"); small[:, 0] = torch.randint(0, 4, (n,), dtype=torch.uint8, device="cuda")
three = uni[:3].repeat((n + 2) // 3, 1)[:n].contiguous()                      # three distinct values, interleaved
w64 = uni.clone(); w64[:, 8:] = 0                                             # 64-bit scalars: 12 of 16 windows empty
top = torch.zeros((n, 32), dtype=torch.uint8, device="cuda"); top[:, 30] = uni[:, 0]; top[:, 31] = uni[:, 1] & 0x1F  # only the top window
# witness-like: 40 % zeros, 30 % ones, 30 % uniform (the shape of many R1CS / PLONK witness columns)
wit = uni.clone()
sel = torch.rand(n, device="cuda")
wit[sel < 0.7] = 0
wit[(sel >= 0.4) & (sel < 0.7), 0] = 1
for name, sc in (("uniform", uni), ("all equal", eq), ("half equal", half), ("2-bit scalars", small), ("3 distinct", three), ("64-bit scalars", w64),
                 ("top window only", top), ("witness-like", wit)):
    ctx.msm(sc)  # (round 5: the first launch of a series that meets a huge bin takes k_sort_fine's fallback and arms k_fine_hist for the next 64 launches: timed is a repeat)
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = ctx.msm(sc); dt = time.perf_counter() - t0
    print("%-14s %8.2f ms  %s" % (name, dt * 1e3, {k: round(v, 3) for k, v in ctx.stage_ms().items()}))
