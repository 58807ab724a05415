#!/bin/bash
# Scratch script of the current GPU-box session (overwritten from session to session; the lasting
# recipes are gpu_round.sh, gpu_ab.sh and gpu_kernel_ab.sh).  Run from the repo root via gpurun.
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
TAG=r04j
timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_hip_api.py tests/test_hip_errors.py -m gpu -q -x > $OUT/${TAG}_tests.log 2>&1; echo "tests rc=$?" >> $OUT/${TAG}_tests.log
tail -4 $OUT/${TAG}_tests.log; grep -E "^(FAILED|ERROR)" $OUT/${TAG}_tests.log | head -20
: > $OUT/AB_${TAG}.jsonl
for V in "" "TDGL_NO_DEFER_CURRENTS=1" "TDGL_GUESS_GRID=512" "" "TDGL_NO_DEFER_CURRENTS=1" "TDGL_GUESS_GRID=512"; do
  env $V timeout 600 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --late-steps 2000 > $OUT/ab_tmp.json 2> $OUT/${TAG}_last.err
  python - "$V" <<'PY' >> $OUT/AB_r04j.jsonl
import json,sys
d=json.load(open('gpurun_out/ab_tmp.json'))
print(json.dumps(dict(variant=sys.argv[1] or "default", head=d["value"], its=d["pcg"]["mean_iterations"], blocked=d["host"]["blocked_frac"], vortex=d["vortex_window"]["value"], vortex_its=d["vortex_window"]["pcg"]["mean_iterations"], sustained=d["sustained"]["value"], sustained_its=d["sustained"]["pcg"]["mean_iterations"], late=d["late_window"]["value"])))
PY
  tail -1 $OUT/AB_r04j.jsonl
done
for WL in 250k strip500k 60k; do
  timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --workload $WL --late-steps 2000 > $OUT/ab_tmp.json 2> $OUT/${TAG}_last.err
  python - "$WL" <<'PY' >> $OUT/AB_r04j.jsonl
import json,sys
d=json.load(open('gpurun_out/ab_tmp.json'))
print(json.dumps(dict(variant=sys.argv[1], head=d["value"], its=d["pcg"]["mean_iterations"], blocked=d["host"]["blocked_frac"], vortex=(d.get("vortex_window") or {}).get("value"), sustained=(d.get("sustained") or {}).get("value"), late=(d.get("late_window") or {}).get("value"), solver=d["setup_s"].get("mu_solver"))))
PY
  tail -1 $OUT/AB_r04j.jsonl
done
