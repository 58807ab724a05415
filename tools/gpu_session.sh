#!/bin/bash
# Scratch script of the current GPU-box session (overwritten from session to session; the lasting
# recipes are gpu_round.sh, gpu_ab.sh and gpu_kernel_ab.sh).  Run from the repo root via gpurun.
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
: > $OUT/AB_r04w.jsonl
for WL in 60k 120k 160k; do
  for MODE in direct amg; do
    if [ $MODE = amg ]; then export TDGL_SUB_MAX_SITES=0 TDGL_DENSE_MAX_SITES=0; else unset TDGL_SUB_MAX_SITES TDGL_DENSE_MAX_SITES; fi
    timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --workload $WL --late-steps 3000 > $OUT/ab_tmp.json 2> $OUT/r04w_last.err
    python - "$WL $MODE" <<'PY' >> $OUT/AB_r04w.jsonl
import json,sys
d=json.load(open('gpurun_out/ab_tmp.json'))
print(json.dumps(dict(variant=sys.argv[1], sites=d["config"]["sites"], head=d["value"], its=d["pcg"]["mean_iterations"], vortex=(d.get("vortex_window") or {}).get("value"), sustained=(d.get("sustained") or {}).get("value"), sustained_its=((d.get("sustained") or {}).get("pcg") or {}).get("mean_iterations"), late=(d.get("late_window") or {}).get("value"), solver=d["setup_s"].get("mu_solver"), setup=d["setup_s"].get("total"))))
PY
    tail -1 $OUT/AB_r04w.jsonl
  done
done
