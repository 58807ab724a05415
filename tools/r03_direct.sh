#!/bin/bash
# direct mu solve for small meshes: its tests, then steps/s with / without it at 5.8k sites (and other small sizes)
OUT=$PWD/gpurun_out
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_hip_direct.py -x -q -m gpu 2>&1 | tail -15
: > $OUT/AB_r03_direct.jsonl
for W in 5k; do
  for LIM in 12288 0; do
    TDGL_DENSE_MAX_SITES=$LIM timeout 600 python bench.py --workload $W --no-cpu-baseline > $OUT/tmp_line.json 2> $OUT/r03_direct.err
    echo "$W dense_max=$LIM rc=$?"; tail -3 $OUT/r03_direct.err
    cat $OUT/tmp_line.json >> $OUT/AB_r03_direct.jsonl
    python - <<'PY'
import json
d=json.load(open('gpurun_out/tmp_line.json'))
print(d['config']['sites'], d['value'], d['ms_per_step'], d['pcg']['mean_iterations'], 'vortex', (d.get('vortex_window') or {}).get('value'), d['host'], d['setup_s'])
PY
  done
done
exit 0
