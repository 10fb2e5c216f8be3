#!/bin/bash
# Is the SMVP's higher cost per addition at 2^22 / 2^24 (67 ps against 60 at 2^20) a clock effect?  Samples the shader clock and power while
# bench.py runs each size (rocm-smi, every ~0.2 s).  usage: bash tools/r04_clock_probe.sh
out=${1:-gpurun_out/r04_clock_probe.txt}
: > $out
for l in 20 24; do
  steps=$([ $l = 20 ] && echo 3000 || echo 150)
  python bench.py --logn $l --steps $steps --warmup 5 --no-cpu-baseline > /tmp/probe_$l.json 2>/dev/null &
  pid=$!
  sleep 6
  for i in $(seq 1 12); do
    kill -0 $pid 2>/dev/null || break
    echo "logn $l sample $i: $(rocm-smi --showclocks --showpower 2>/dev/null | grep -i -E 'sclk|Power \(W\)|Socket Power' | tr -s ' ' | tr '\n' ';')" >> $out
    sleep 0.2
  done
  wait $pid
  python - $l >> $out <<'PY'
import json,sys
d=json.loads([l for l in open('/tmp/probe_%s.json' % sys.argv[1]) if l.startswith('{')][-1])
print('logn %s: ms/step %.4f smvp_ms %.4f' % (sys.argv[1], d['ms_per_step'], d['roofline']['kernel_ms']))
PY
done
cat $out
