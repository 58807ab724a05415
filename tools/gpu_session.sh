#!/bin/bash
# Scratch script of the current GPU-box session (overwritten from session to session; the lasting
# recipes are gpu_round.sh, gpu_ab.sh and gpu_kernel_ab.sh).  Run from the repo root via gpurun.
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
: > $OUT/SOAK_r04.jsonl
for JOB in "1M 40000" "250k 100000" "strip500k 60000"; do
  echo "{\"soak\": \"$JOB\"}" >> $OUT/SOAK_r04.jsonl
  timeout 900 python tools/soak.py $JOB >> $OUT/SOAK_r04.jsonl 2> $OUT/soak_last.err; echo "$JOB rc=$?"
done
echo "{\"soak\": \"60k forced AMG-PCG 150000\"}" >> $OUT/SOAK_r04.jsonl
TDGL_SUB_MAX_SITES=0 TDGL_DENSE_MAX_SITES=0 timeout 900 python tools/soak.py 60k 150000 >> $OUT/SOAK_r04.jsonl 2> $OUT/soak_last.err; echo "60k rc=$?"
python - <<'PY'
import json
for l in open("gpurun_out/SOAK_r04.jsonl"):
    d=json.loads(l)
    if "soak" in d: print(d); continue
    last=d
    if d["steps"] % 20000 == 0 or d["steps"] <= 2500: print(d["steps"], d["wall_s"], d["time"], d["dt_last"], d["pcg_mean"], d["pcg_max"], d["psi_retries"], d["fp64_fallbacks"], d["finite"], d["sites_below_0p1"])
PY
