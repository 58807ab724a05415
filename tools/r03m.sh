#!/bin/bash
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_hip_errors.py tests/test_hip_api.py -m gpu -q -x > $OUT/r03m_tests.log 2>&1; tail -3 $OUT/r03m_tests.log
: > $OUT/AB_r03m.jsonl
for W in 5k 60k; do
 for E in "" "TDGL_NO_SMALL_PCG=1"; do
  timeout 900 env $E python bench.py --workload $W --no-cpu-baseline --vortex-window off > $OUT/ab_tmp.json 2> $OUT/ab_r03m_last.err || tail -3 $OUT/ab_r03m_last.err
  python - "$W $E" <<'PY' >> $OUT/AB_r03m.jsonl
import json,sys
d=json.load(open('gpurun_out/ab_tmp.json'))
print(json.dumps(dict(variant=sys.argv[1], value=d['value'], ms=d['ms_per_step'], its=d['pcg']['mean_iterations'], levels=d['config']['amg_levels'], host=d['host'])))
PY
  tail -1 $OUT/AB_r03m.jsonl
 done
done
