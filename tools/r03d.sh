#!/bin/bash
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 2700 python -m pytest tests -m gpu -q > $OUT/r03d_tests.log 2>&1; echo "tests rc=$?" >> $OUT/r03d_tests.log; tail -3 $OUT/r03d_tests.log; grep -E "^(FAILED|ERROR)" $OUT/r03d_tests.log | head
: > $OUT/AB_r03d.jsonl
for W in 5k 60k 250k 1M; do
 for E in "" "TDGL_FULL_PARTIALS=1"; do
  timeout 900 env $E python bench.py --workload $W --no-cpu-baseline --vortex-window off > $OUT/ab_tmp.json 2> $OUT/ab_r03d_last.err || tail -3 $OUT/ab_r03d_last.err
  python - "$W $E" <<'PY' >> $OUT/AB_r03d.jsonl
import json,sys
d=json.load(open('gpurun_out/ab_tmp.json'))
print(json.dumps(dict(variant=sys.argv[1], value=d['value'], ms=d['ms_per_step'], its=d['pcg']['mean_iterations'])))
PY
  tail -1 $OUT/AB_r03d.jsonl
 done
done
