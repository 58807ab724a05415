#!/bin/bash
bash tools/gpu_round.sh r03i tests pmc
OUT=$PWD/gpurun_out
for W in 5k 60k 250k strip500k strip500k_ff 4M; do
  timeout 1200 python bench.py --workload $W --no-cpu-baseline --trace-iterations $OUT/r03i_trace_$W.json > $OUT/BENCH_r03i_$W.json 2> $OUT/r03i_$W.err
  echo "$W rc=$?"; python - $W <<'PY'
import json,sys
d=json.load(open(f'gpurun_out/BENCH_r03i_{sys.argv[1]}.json'))
print(d['value'], d['ms_per_step'], d['pcg']['mean_iterations'], 'vortex', (d.get('vortex_window') or {}).get('value'), d['host'], d['setup_s'])
PY
done
exit 0
