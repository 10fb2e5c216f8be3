#!/bin/bash
# Round-3 A/B set for the window-sharded share of an 8-rank run on ONE GPU (BENCH_EMULATE_WORLD: a tuning aid, never a reported result)
# and the single-GPU line.  Usage (on the GPU box): bash tools/r03_share_ab.sh <tag>   -> gpurun_out/<tag>_*.json
set -e
tag=${1:-r03}
out=gpurun_out
mkdir -p $out
run() { name=$1; shift; echo "== $name"; env "$@" python bench.py --no-cpu-baseline ${STEPS_ARGS} > $out/${tag}_${name}.json 2> $out/${tag}_${name}.err || { tail -5 $out/${tag}_${name}.err; return 1; }; python - $out/${tag}_${name}.json <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("   value %.1f MSM/s  ms/step %.4f  cold %.4f  smvp_ms %.4f  frac %.4f  group %s" % (d["value"], d["ms_per_step"], d.get("ms_per_step_cold_protocol", 0), d["roofline"]["kernel_ms"], d["roofline"]["frac"], d["config"]["msms_per_launch"]))
PY
}
for steps in 20 64; do
  STEPS_ARGS="--steps $steps --warmup 5"
  run single_endo_s$steps BENCH_BASES=endomorphism
  run share8_endo_s$steps BENCH_EMULATE_WORLD=8 BENCH_BASES=endomorphism
  run share8_plain_s$steps BENCH_EMULATE_WORLD=8 BENCH_BASES=plain
  run share8_plain_noplanes_s$steps BENCH_EMULATE_WORLD=8 BENCH_BASES=plain MSM_HIP_PLANES_MAX_W=0
  run share8_endo_noplanes_s$steps BENCH_EMULATE_WORLD=8 BENCH_BASES=endomorphism MSM_HIP_PLANES_MAX_W=0
done
STEPS_ARGS="--steps 20 --warmup 5"
run share8_endo_s20_g8 BENCH_EMULATE_WORLD=8 BENCH_BASES=endomorphism BENCH_MSMS_PER_LAUNCH=8
run share8_plain_s20_g8 BENCH_EMULATE_WORLD=8 BENCH_BASES=plain BENCH_MSMS_PER_LAUNCH=8
run share4_endo_s20 BENCH_EMULATE_WORLD=4 BENCH_BASES=endomorphism
run share2_endo_s20 BENCH_EMULATE_WORLD=2 BENCH_BASES=endomorphism
run single_plain_s20 BENCH_BASES=plain
run single_endo_planes_whole_s20 BENCH_BASES=endomorphism MSM_HIP_PLANES_WHOLE=1
run single_plain_planes_whole_s20 BENCH_BASES=plain MSM_HIP_PLANES_WHOLE=1
echo "== native mgpu, 8 contexts on GPU 0"
BENCH_MGPU_NATIVE=1 BENCH_MGPU_IDS=0,0,0,0,0,0,0,0 python bench.py --gpus 8 --steps 20 --warmup 5 > $out/${tag}_native_mgpu_8ctx_s20.json 2> $out/${tag}_native_mgpu_8ctx_s20.err || tail -5 $out/${tag}_native_mgpu_8ctx_s20.err
tail -c 900 $out/${tag}_native_mgpu_8ctx_s20.json
