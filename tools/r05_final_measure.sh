#!/bin/bash
# Round-5 measurement set on the final code (run on the GPU box from the repo root):  bash tools/r05_final_measure.sh <tag> bench|prof|all
#   bench: the driver's command three times (the spread of `value`, `value_steady_state` and the wide-tables line on ONE box: VERDICT r04 item 4),
#          the other sizes, the other base modes, one rank's share of 8 / 4 / 2 with plain and wide-table shares, the other curves.
#   prof : rocprofv3 --kernel-trace --stats and the separate --pmc passes of the default bench at 2^20 and 2^24, and of one rank's share of 8.
# Two gpurun calls: the whole set does not fit one call's time limit.
tag=${1:-r05_final}
part=${2:-all}
out=gpurun_out
mkdir -p $out
show() { python - $1 <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
w=d.get("fixed_base_tables_wide") or {}
print("   %s: value %.1f  steady %.1f  wide %.1f  ms/step %.4f  smvp_ms %.4f  frac %.4f  lat %s" % (sys.argv[1].split("/")[-1], d["value"], d.get("value_steady_state") or 0, w.get("value", 0), d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"], d.get("latency_ms_single_msm")))
PY
}
if [ $part != prof ]; then
for r in 1 2 3; do
  python bench.py --steps 20 --warmup 5 > $out/${tag}_bench_run$r.json 2> $out/${tag}_bench_run$r.err && show $out/${tag}_bench_run$r.json
done
cp $out/${tag}_bench_run1.json $out/${tag}_bench.json
for l in 14 16 18 19 22 24; do python bench.py --steps 20 --warmup 5 --logn $l --no-cpu-baseline > $out/${tag}_bench_logn$l.json 2>/dev/null && show $out/${tag}_bench_logn$l.json; done
for b in plain tables tables_wide; do
  BENCH_BASES=$b BENCH_CONFIGS=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/${tag}_bench_${b}_bases.json 2>/dev/null && show $out/${tag}_bench_${b}_bases.json
done
for w in 8 4 2; do
  BENCH_EMULATE_WORLD=$w python bench.py --steps 64 --warmup 8 --no-cpu-baseline > $out/${tag}_bench_emulated_share_of_$w.json 2>/dev/null && show $out/${tag}_bench_emulated_share_of_$w.json
  BENCH_EMULATE_WORLD=$w BENCH_BASES=tables_wide python bench.py --steps 64 --warmup 8 --no-cpu-baseline > $out/${tag}_bench_emulated_share_of_${w}_wide.json 2>/dev/null && show $out/${tag}_bench_emulated_share_of_${w}_wide.json
done
BENCH_EMULATE_WORLD=8 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/${tag}_bench_emulated_share_of_8_steps20.json 2>/dev/null && show $out/${tag}_bench_emulated_share_of_8_steps20.json
BENCH_EMULATE_WORLD=8 BENCH_BASES=tables_wide python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/${tag}_bench_emulated_share_of_8_wide_steps20.json 2>/dev/null && show $out/${tag}_bench_emulated_share_of_8_wide_steps20.json
timeout -k 10 400 python tools/curve_throughput.py 20 grumpkin pallas bls12_381 bn254_g2 bls12_381_g2 > $out/${tag}_curves_throughput.txt 2>&1; grep -v amdgpu.ids $out/${tag}_curves_throughput.txt
fi
if [ $part = bench ]; then exit 0; fi
bash tools/profile_round.sh $tag 20 > $out/${tag}_prof.log 2>&1; tail -2 $out/${tag}_prof.log
python tools/pmc_summarize.py $out/prof_$tag 20 8 $out/${tag}_smvp_pmc_traffic.json endomorphism > $out/${tag}_pmc_summary.txt 2>&1; tail -16 $out/${tag}_pmc_summary.txt
bash tools/profile_round.sh ${tag}_logn24 24 > $out/${tag}_prof24.log 2>&1; tail -2 $out/${tag}_prof24.log
python tools/pmc_summarize.py $out/prof_${tag}_logn24 24 8 $out/${tag}_smvp_pmc_traffic_logn24.json endomorphism > $out/${tag}_logn24_pmc_summary.txt 2>&1; tail -16 $out/${tag}_logn24_pmc_summary.txt
# kernel stats of one rank's share of an 8-rank run, plain and wide-table shares (the kernels of the window-sharded path)
cd /tmp && export TMPDIR=/tmp
BENCH_EMULATE_WORLD=8 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof_${tag}_share8 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 16 --warmup 8 --no-cpu-baseline > $GRAFT_REPO_ROOT/$out/${tag}_share8_trace.log 2>&1
BENCH_EMULATE_WORLD=8 BENCH_BASES=tables_wide timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof_${tag}_share8_wide -- python3 $GRAFT_REPO_ROOT/bench.py --steps 16 --warmup 8 --no-cpu-baseline > $GRAFT_REPO_ROOT/$out/${tag}_share8_wide_trace.log 2>&1
find $GRAFT_REPO_ROOT/$out/prof_${tag}_share8 $GRAFT_REPO_ROOT/$out/prof_${tag}_share8_wide -name "*kernel_stats.csv" | head -4
