#!/bin/bash
# dense inverse against the substructured solve under the run-ahead loop, 2k-5.8k sites
OUT=$PWD/gpurun_out
: > $OUT/AB_r03_sub5.jsonl
run() {
    env $2 timeout 120 python bench.py --workload $1 --no-cpu-baseline --vortex-window off --steps 2000 --warmup 200 > $OUT/tmp_line.json 2> $OUT/r03_sub.err
    echo "$1 $2 rc=$?"
    cat $OUT/tmp_line.json >> $OUT/AB_r03_sub5.jsonl
    python - <<'PY'
import json
try:
    d=json.load(open('gpurun_out/tmp_line.json'))
    print('   ', d['config']['sites'], d['value'], 'steps/s', d['ms_per_step'], 'ms', d['host'], d['setup_s'].get('mu_solver','')[:24], (d['setup_s'].get('substructure') or {}).get('parts'))
except Exception as e: print('   no line', e)
PY
}
for W in 2k 4k 5k; do
  run $W "TDGL_DENSE_MAX_SITES=20000"
  for B in 128 192 256; do run $W "TDGL_DENSE_MAX_SITES=500 TDGL_SUB_BLOCK=$B"; done
done
run 9k "TDGL_SUB_BLOCK=192"
run 9k "TDGL_SUB_BLOCK=256"
run 23k "TDGL_SUB_BLOCK=256"
exit 0
