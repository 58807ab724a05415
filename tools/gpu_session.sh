#!/bin/bash
# Scratch script of the current GPU-box session (overwritten from session to session; the lasting
# recipes are gpu_round.sh, gpu_ab.sh and gpu_kernel_ab.sh).  Run from the repo root via gpurun.
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
: > $OUT/BENCH_r04_parity_lines.jsonl
for WL in 5k 60k 250k strip500k strip500k_ff; do
  timeout 900 python bench.py --steps 20 --warmup 5 --workload $WL --late-steps 2000 >> $OUT/BENCH_r04_parity_lines.jsonl 2> $OUT/r04q_$WL.err
  echo "$WL rc=$?"
done
python - <<'PY'
import json
for l in open("gpurun_out/BENCH_r04_parity_lines.jsonl"):
    try: d = json.loads(l)
    except Exception: continue
    def par(x):
        p = (x or {}).get("parity_vs_oracle")
        return None if not p else max(p[k] for k in ("dt","abs_sq_psi","mu_zero_mean","J_s","J_n"))
    def w(x):
        return None if not x else (x["value"], x["pcg"]["mean_iterations"], par(x))
    print(d["config"]["workload"][:40], "| head", d["value"], d["pcg"]["mean_iterations"], par(d), "| vortex", w(d.get("vortex_window")), "| sustained", (d.get("sustained") or {}).get("value"), "| late", w(d.get("late_window")), "| cpu", (d.get("cpu_baseline") or {}).get("value"))
PY
