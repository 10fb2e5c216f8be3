#!/bin/bash
# Round 4: where the wide fixed-base mode's time goes (run on the GPU box from the repo root): kernel stats of bench.py in the two table modes,
# and the wide mode at other lane counts.
out=gpurun_out
mkdir -p $out
show() { python - $1 "$2" <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("%s: value %.1f  ms/step %.4f  smvp_ms %.4f  lat %s" % (sys.argv[2], d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d.get("latency_ms_single_msm")))
PY
}
for lanes in 589824 442368 393216 786432; do
  BENCH_BASES=tables_wide MSM_HIP_TARGET_LANES=$lanes python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/wide_lanes_$lanes.json 2>/dev/null && show $out/wide_lanes_$lanes.json "tables_wide lanes $lanes"
done
cd /tmp && export TMPDIR=/tmp
for mode in tables_wide tables endomorphism; do
  export BENCH_BASES=$mode
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof_wide_$mode -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $GRAFT_REPO_ROOT/$out/wide_trace_$mode.log 2>&1
  f=$(find $GRAFT_REPO_ROOT/$out/prof_wide_$mode -name "*kernel_stats.csv" | head -1)
  echo "== $mode"; python3 - $f <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print("%-60s calls %6s avg_us %10.1f total_ms %9.2f" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
done
