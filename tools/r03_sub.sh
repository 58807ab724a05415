#!/bin/bash
# substructured direct solve with the factors built on the device: tests, steps/s over sizes and block sizes
OUT=$PWD/gpurun_out
mkdir -p $OUT
timeout 900 python -m pytest tests/test_hip_direct.py -x -q -m gpu -k "substructured or agree" 2>&1 | tail -8
TDGL_SUB_HOST=1 timeout 900 python -m pytest tests/test_hip_direct.py -x -q -m gpu -k "substructured_solve_matches" 2>&1 | tail -3
: > $OUT/AB_r03_sub2.jsonl
run() {
    env $2 timeout 600 python bench.py --workload $1 --no-cpu-baseline --vortex-window off --steps 1000 --warmup 100 > $OUT/tmp_line.json 2> $OUT/r03_sub.err
    echo "$1 $2 rc=$?"; grep "set-up" $OUT/r03_sub.err | cut -c1-400
    cat $OUT/tmp_line.json >> $OUT/AB_r03_sub2.jsonl
    python - <<'PY'
import json
try:
    d=json.load(open('gpurun_out/tmp_line.json'))
    print('   ', d['config']['sites'], d['value'], 'steps/s', d['ms_per_step'], 'ms', d['pcg']['mean_iterations'], 'it', d['host'])
except Exception as e: print('   no line', e)
PY
}
S="TDGL_DENSE_MAX_SITES=5000 TDGL_SUB_MAX_SITES=400000"
for W in 9k 23k 60k; do
  for B in 320 448 640; do run $W "$S TDGL_SUB_BLOCK=$B"; done
done
run 250k "$S TDGL_SUB_BLOCK=448"
run 250k "$S TDGL_SUB_BLOCK=1024"
run 250k "TDGL_DENSE_MAX_SITES=0 TDGL_SUB_MAX_SITES=0"
bash tools/gpu_kernel_ab.sh r03_sub2 "$S@--workload 60k"
exit 0
