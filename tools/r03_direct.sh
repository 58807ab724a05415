#!/bin/bash
# direct mu solve for small meshes: its tests, steps/s with the device-built / host-built inverse / AMG-PCG at
# several sizes, kernel traces at 5.8k sites
OUT=$PWD/gpurun_out
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_hip_direct.py -x -q -m gpu 2>&1 | tail -5
TDGL_DENSE_HOST=1 timeout 1500 python -m pytest tests/test_hip_direct.py -x -q -m gpu -k "lu or agree or retries" 2>&1 | tail -3
: > $OUT/AB_r03_direct2.jsonl
for W in 5k 2k 9k 12k 16k; do
  for V in "TDGL_DENSE_MAX_SITES=30000" "TDGL_DENSE_MAX_SITES=30000 TDGL_DENSE_HOST=1" "TDGL_DENSE_MAX_SITES=0"; do
    env $V timeout 600 python bench.py --workload $W --no-cpu-baseline --vortex-window off > $OUT/tmp_line.json 2> $OUT/r03_direct.err
    echo "$W $V rc=$?"
    cat $OUT/tmp_line.json >> $OUT/AB_r03_direct2.jsonl
    python - <<'PY'
import json
d=json.load(open('gpurun_out/tmp_line.json'))
print('   ', d['config']['sites'], d['value'], 'steps/s', d['ms_per_step'], 'ms', d['pcg']['mean_iterations'], 'it', d['host'], d['setup_s'])
PY
  done
done
bash tools/gpu_kernel_ab.sh r03_direct2 "TDGL_DENSE_MAX_SITES=30000@--workload 5k" "TDGL_DENSE_MAX_SITES=30000@--workload 2k"
exit 0
