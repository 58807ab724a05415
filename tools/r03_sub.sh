#!/bin/bash
# substructured direct solve: tests, steps/s over sizes and block sizes against AMG-PCG, kernel trace at 59k sites
OUT=$PWD/gpurun_out
mkdir -p $OUT
timeout 900 python -m pytest tests/test_hip_direct.py -x -q -m gpu 2>&1 | tail -4
: > $OUT/AB_r03_sub3.jsonl
run() {
    env $2 timeout 600 python bench.py --workload $1 --no-cpu-baseline --vortex-window off --steps 1000 --warmup 100 > $OUT/tmp_line.json 2> $OUT/r03_sub.err
    echo "$1 $2 rc=$?"
    cat $OUT/tmp_line.json >> $OUT/AB_r03_sub3.jsonl
    python - <<'PY'
import json
try:
    d=json.load(open('gpurun_out/tmp_line.json'))
    print('   ', d['config']['sites'], d['value'], 'steps/s', d['ms_per_step'], 'ms', d['pcg']['mean_iterations'], 'it', d['host'], d['setup_s'])
except Exception as e: print('   no line', e)
PY
}
S="TDGL_SUB_MAX_SITES=400000"
for W in 9k 23k 60k; do
  for B in 256 320; do run $W "$S TDGL_SUB_BLOCK=$B"; done
done
for W in 120k 160k; do
  for B in 320 512; do run $W "$S TDGL_SUB_BLOCK=$B"; done
  run $W "TDGL_DENSE_MAX_SITES=0 TDGL_SUB_MAX_SITES=0"
done
bash tools/gpu_kernel_ab.sh r03_sub3 "$S@--workload 60k" "$S@--workload 23k"
exit 0
