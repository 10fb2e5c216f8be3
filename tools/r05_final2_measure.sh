#!/bin/bash
# After the late change of k_smvp_stitch (sorted lane assignment): the driver's command three times, 2^16, one rank's share of 8 (plain / wide), then the
# rocprofv3 passes of tools/r05_final_measure.sh again, so that profiles/rocprof_kernel_ms.json and the PMC traffic describe the code as committed.
#   bash tools/r05_final2_measure.sh [tag]
tag=${1:-r05_final2}
out=gpurun_out
mkdir -p $out
show() { python - $1 <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
w=d.get("fixed_base_tables_wide") or {}
print("   %s: value %.1f  steady %.1f  wide %.1f  ms/step %.4f  smvp_ms %.4f  frac %.4f  lat %s" % (sys.argv[1].split("/")[-1], d["value"], d.get("value_steady_state") or 0, w.get("value", 0), d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"], d.get("latency_ms_single_msm")))
PY
}
for r in 1 2 3; do
  python bench.py --steps 20 --warmup 5 > $out/${tag}_bench_run$r.json 2> $out/${tag}_bench_run$r.err && show $out/${tag}_bench_run$r.json
done
python bench.py --steps 20 --warmup 5 --logn 16 --no-cpu-baseline > $out/${tag}_bench_logn16.json 2>/dev/null && show $out/${tag}_bench_logn16.json
BENCH_EMULATE_WORLD=8 python bench.py --steps 64 --warmup 8 --no-cpu-baseline > $out/${tag}_bench_emulated_share_of_8.json 2>/dev/null && show $out/${tag}_bench_emulated_share_of_8.json
BENCH_EMULATE_WORLD=8 BENCH_BASES=tables_wide python bench.py --steps 64 --warmup 8 --no-cpu-baseline > $out/${tag}_bench_emulated_share_of_8_wide.json 2>/dev/null && show $out/${tag}_bench_emulated_share_of_8_wide.json
bash tools/r05_final_measure.sh $tag prof
