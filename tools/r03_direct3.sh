#!/bin/bash
# direct solve with the deferred edge currents: its tests, steps/s, then the whole GPU suite
OUT=$PWD/gpurun_out
mkdir -p $OUT
timeout 900 python -m pytest tests/test_hip_direct.py -x -q -m gpu 2>&1 | tail -5
: > $OUT/AB_r03_direct3.jsonl
for W in 5k 2k 12k 16k; do
    timeout 600 python bench.py --workload $W --no-cpu-baseline --vortex-window off --steps 2000 --warmup 200 > $OUT/tmp_line.json 2> $OUT/r03_direct.err
    echo "$W rc=$?"
    cat $OUT/tmp_line.json >> $OUT/AB_r03_direct3.jsonl
    python - <<'PY'
import json
d=json.load(open('gpurun_out/tmp_line.json'))
print('   ', d['config']['sites'], d['value'], 'steps/s', d['ms_per_step'], 'ms', d['host'], d['setup_s'])
PY
done
bash tools/gpu_kernel_ab.sh r03_direct3 "A=1@--workload 5k"
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -8
exit 0
