#!/bin/bash
# round 3, second GPU session: full suite, bench + kernel trace + PMC of the current code, flexible vs FR beta, strips
bash tools/gpu_round.sh r03b tests pmc
bash tools/gpu_ab.sh r03b "" "--cg-flexible" "" "--cg-flexible"
OUT=$PWD/gpurun_out
for W in "strip500k_ps" "strip500k_ff" "5k" "60k" "250k"; do
  timeout 900 python bench.py --workload $W --no-cpu-baseline --trace-iterations $OUT/r03b_trace_$W.json > $OUT/BENCH_r03b_$W.json 2> $OUT/r03b_$W.err
  echo "$W rc=$?"; python - $W <<'PY'
import json,sys
d=json.load(open(f'gpurun_out/BENCH_r03b_{sys.argv[1]}.json'))
for k in ('value','ms_per_step','vortex_window','host','setup_s'): print(k, d.get(k))
PY
done
exit 0
