#!/bin/bash
# Run on the GPU box from the repo root:  bash tools/profile_round.sh <tag> [logn]
# Produces, under gpurun_out/prof_<tag>/: kernel-trace stats of bench.py, FETCH_SIZE and WRITE_SIZE counter passes
# (separate runs, as MI355X_MICROARCH.md prescribes), and the same counters on the calibration kernels.
set -o pipefail
TAG=${1:-r01}; LOGN=${2:-20}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT && cd /tmp && export TMPDIR=/tmp
hipcc -O3 --offload-arch=gfx950 $ROOT/tools/pmc_calibrate.hip -o /tmp/pmc_calibrate || exit 1
BENCH="python3 $ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --logn $LOGN"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/trace.log 2>&1 || exit 2
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $BENCH > $OUT/pmc_fetch.log 2>&1 || exit 3
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $BENCH > $OUT/pmc_write.log 2>&1 || exit 4
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/cal_fetch -- /tmp/pmc_calibrate > $OUT/cal_fetch.log 2>&1 || exit 5
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/cal_write -- /tmp/pmc_calibrate > $OUT/cal_write.log 2>&1 || exit 6
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -- $BENCH > $OUT/pmc_sq.log 2>&1 || echo "sq pass failed (non-fatal)"
find $OUT -name "*.csv" | head -40
