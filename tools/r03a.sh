#!/bin/bash
# round 3, first GPU session: full GPU suite, headline bench with parity + vortex window, strips with/without probes
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/r03a_tests.log 2>&1; echo "tests rc=$?" >> $OUT/r03a_tests.log; tail -3 $OUT/r03a_tests.log
timeout 1200 python bench.py --steps 20 --warmup 5 --trace-iterations $OUT/r03a_trace_1M.json > $OUT/BENCH_r03a_1M_driver.json 2> $OUT/r03a_bench_driver.err
echo "bench(driver flags) rc=$?"; python - <<'PY'
import json
d=json.load(open('gpurun_out/BENCH_r03a_1M_driver.json'))
for k in ('value','ms_per_step','parity_vs_oracle','vortex_window','host','setup_s','cpu_baseline'): print(k, d.get(k))
PY
for W in "strip500k" "strip500k --no-probes" "strip500k_ps"; do
  TAG=$(echo $W | tr -d ' -'); 
  timeout 900 python bench.py --workload $W --no-cpu-baseline --trace-iterations $OUT/r03a_trace_$TAG.json > $OUT/BENCH_r03a_$TAG.json 2> $OUT/r03a_$TAG.err
  echo "$W rc=$?"; python - $TAG <<'PY'
import json,sys
d=json.load(open(f'gpurun_out/BENCH_r03a_{sys.argv[1]}.json'))
for k in ('value','ms_per_step','vortex_window','host'): print(k, d.get(k))
PY
done
exit 0
