#!/bin/bash
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q -k "graded or very_small" 2>&1 | tail -3
: > $OUT/AB_r03e.jsonl
for W in 5k 60k 250k 1M; do
 for E in "" "TDGL_NO_GRAPH=1"; do
  timeout 900 env $E python bench.py --workload $W --no-cpu-baseline --vortex-window off > $OUT/ab_tmp.json 2> $OUT/ab_r03e_last.err || tail -3 $OUT/ab_r03e_last.err
  python - "$W $E" <<'PY' >> $OUT/AB_r03e.jsonl
import json,sys
d=json.load(open('gpurun_out/ab_tmp.json'))
print(json.dumps(dict(variant=sys.argv[1], value=d['value'], ms=d['ms_per_step'], its=d['pcg']['mean_iterations'], host=d['host'])))
PY
  tail -1 $OUT/AB_r03e.jsonl
 done
done
