#!/bin/bash
# run-ahead time loop: tests (direct suite + the full suite), steps/s with / without it over sizes
OUT=$PWD/gpurun_out
mkdir -p $OUT
timeout 300 python -m pytest tests/test_hip_direct.py -x -q -m gpu 2>&1 | tail -3
: > $OUT/AB_r03_ra.jsonl
run() {
    env $2 timeout 150 python bench.py --workload $1 --no-cpu-baseline --steps 2000 --warmup 200 > $OUT/tmp_line.json 2> $OUT/r03_ra.err
    echo "$1 $2 rc=$?"; tail -1 $OUT/r03_ra.err | cut -c1-200
    cat $OUT/tmp_line.json >> $OUT/AB_r03_ra.jsonl
    python - <<'PY'
import json
try:
    d=json.load(open('gpurun_out/tmp_line.json'))
    print('   ', d['config']['sites'], d['value'], 'steps/s', d['ms_per_step'], 'ms', d['host'], 'vortex', (d.get('vortex_window') or {}).get('value'), 'late', (d.get('late_window') or {}).get('value'), (d.get('late_window') or {}).get('retries'))
except Exception as e: print('   no line', e)
PY
}
for W in 2k 5k 9k 23k 60k 120k; do
  run $W "A=1"
  run $W "TDGL_NO_RUN_AHEAD=1"
done
exit 0
