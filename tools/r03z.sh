#!/bin/bash
# final GPU session of round 3: suite, bench lines, kernel trace + PMC, the other workloads, screening kernel, microbenchmarks
bash tools/gpu_round.sh r03z tests pmc
OUT=$PWD/gpurun_out
: > $OUT/BENCH_r03z_other_workloads.jsonl
for W in 5k 60k 250k strip500k "strip500k --no-probes" strip500k_ff 4M; do
  timeout 1200 python bench.py --workload $W --no-cpu-baseline > $OUT/tmp_line.json 2> $OUT/r03z_other.err
  echo "$W rc=$?"; cat $OUT/tmp_line.json >> $OUT/BENCH_r03z_other_workloads.jsonl
  python - <<'PY'
import json
d=json.load(open('gpurun_out/tmp_line.json'))
print(d['config']['sites'], d['value'], d['ms_per_step'], d['pcg']['mean_iterations'], 'vortex', (d.get('vortex_window') or {}).get('value'), d['host'], d['setup_s'])
PY
done
timeout 600 python tools/bench_screening.py > $OUT/SCREENING_r03z.jsonl 2> $OUT/r03z_screening.err; tail -3 $OUT/SCREENING_r03z.jsonl
timeout 300 python tools/bench_barrier.py > $OUT/r03z_microbench.txt 2>&1; cat $OUT/r03z_microbench.txt
exit 0
