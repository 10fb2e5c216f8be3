#!/bin/bash
# Round-4 measurement set (one set per round: VERDICT r03 item 8) (run on the GPU box from the repo root):  bash tools/r04_final_measure.sh <tag>
# bench lines at every size, the emulated per-rank shares, the plain-bases line, and the rocprofv3 passes at 2^20 and 2^24.
tag=${1:-r04_v1}
part=${2:-all}   # bench | prof | all  (two gpurun calls: the whole set does not fit one call's time limit)
out=gpurun_out
mkdir -p $out
show() { python - $1 <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("   %s: value %.1f  ms/step %.4f  cold %.4f  smvp_ms %.4f  frac %.4f  lat %s" % (sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], d.get("ms_per_step_cold_protocol", 0), d["roofline"]["kernel_ms"], d["roofline"]["frac"], d.get("latency_ms_single_msm")))
PY
}
if [ $part != prof ]; then
python bench.py --steps 20 --warmup 5 > $out/${tag}_bench.json 2> $out/${tag}_bench.err && show $out/${tag}_bench.json
for l in 14 16 18 19 22 24; do python bench.py --steps 20 --warmup 5 --logn $l --no-cpu-baseline > $out/${tag}_bench_logn$l.json 2>/dev/null && show $out/${tag}_bench_logn$l.json; done
BENCH_BASES=plain python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/${tag}_bench_plain_bases.json 2>/dev/null && show $out/${tag}_bench_plain_bases.json
BENCH_BASES=tables python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/${tag}_bench_tables.json 2>/dev/null && show $out/${tag}_bench_tables.json
timeout -k 10 300 python tools/curve_throughput.py 20 grumpkin pallas bls12_381 > $out/${tag}_curves_throughput.txt 2>&1; grep -v amdgpu.ids $out/${tag}_curves_throughput.txt
for w in 8 4 2; do
  BENCH_EMULATE_WORLD=$w python bench.py --steps 64 --warmup 8 --no-cpu-baseline > $out/${tag}_bench_emulated_share_of_$w.json 2>/dev/null && show $out/${tag}_bench_emulated_share_of_$w.json
  BENCH_EMULATE_WORLD=$w python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/${tag}_bench_emulated_share_of_${w}_steps20.json 2>/dev/null && show $out/${tag}_bench_emulated_share_of_${w}_steps20.json
done
BENCH_EMULATE_WORLD=8 BENCH_BASES=endomorphism python bench.py --steps 64 --warmup 8 --no-cpu-baseline > $out/${tag}_bench_emulated_share_of_8_half_windows.json 2>/dev/null && show $out/${tag}_bench_emulated_share_of_8_half_windows.json
fi
if [ $part = bench ]; then exit 0; fi
bash tools/profile_round.sh $tag 20 > $out/${tag}_prof.log 2>&1; tail -2 $out/${tag}_prof.log
python tools/pmc_summarize.py $out/prof_$tag 20 8 $out/${tag}_smvp_pmc_traffic.json endomorphism > $out/${tag}_pmc_summary.txt 2>&1; tail -16 $out/${tag}_pmc_summary.txt
bash tools/profile_round.sh ${tag}_logn24 24 > $out/${tag}_prof24.log 2>&1; tail -2 $out/${tag}_prof24.log
python tools/pmc_summarize.py $out/prof_${tag}_logn24 24 8 $out/${tag}_smvp_pmc_traffic_logn24.json endomorphism > $out/${tag}_logn24_pmc_summary.txt 2>&1; tail -16 $out/${tag}_logn24_pmc_summary.txt
# kernel stats of one rank's share of an 8-rank run (the kernels of the window-sharded path: k_scatter_planes, ...)
cd /tmp && export TMPDIR=/tmp
BENCH_EMULATE_WORLD=8 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof_${tag}_share8 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 16 --warmup 8 --no-cpu-baseline > $GRAFT_REPO_ROOT/$out/${tag}_share8_trace.log 2>&1
find $GRAFT_REPO_ROOT/$out/prof_${tag}_share8 -name "*kernel_stats.csv" | head -2
