#!/bin/bash
# where the substructured solve overtakes the dense inverse (2k-9k sites)
OUT=$PWD/gpurun_out
: > $OUT/AB_r03_sub4.jsonl
run() {
    env $2 timeout 600 python bench.py --workload $1 --no-cpu-baseline --vortex-window off --steps 2000 --warmup 200 > $OUT/tmp_line.json 2> $OUT/r03_sub.err
    echo "$1 $2 rc=$?"
    cat $OUT/tmp_line.json >> $OUT/AB_r03_sub4.jsonl
    python - <<'PY'
import json
try:
    d=json.load(open('gpurun_out/tmp_line.json'))
    print('   ', d['config']['sites'], d['value'], 'steps/s', d['ms_per_step'], 'ms', d['host'], d['setup_s'])
except Exception as e: print('   no line', e)
PY
}
for W in 2k 5k 9k; do
  run $W "TDGL_DENSE_MAX_SITES=20000"
  for B in 128 192 320; do run $W "TDGL_DENSE_MAX_SITES=500 TDGL_SUB_BLOCK=$B"; done
done
run 120k ""
run 160k "TDGL_SUB_MAX_SITES=400000"
exit 0
