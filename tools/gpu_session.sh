#!/bin/bash
# Scratch script of the current GPU-box session (overwritten from session to session; the lasting
# recipes are gpu_round.sh, gpu_ab.sh and gpu_kernel_ab.sh).  Run from the repo root via gpurun.
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
TAG=r04a
timeout 2400 python -m pytest tests -m gpu -q -x > $OUT/${TAG}_tests.log 2>&1; echo "tests rc=$?" >> $OUT/${TAG}_tests.log
tail -5 $OUT/${TAG}_tests.log; grep -E "^(FAILED|ERROR)" $OUT/${TAG}_tests.log | head
: > $OUT/AB_${TAG}.jsonl
for W in 6 12 16; do
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --guess-window $W >> $OUT/AB_${TAG}.jsonl 2> $OUT/${TAG}_w$W.err
  echo "window $W rc=$?"
done
for R in 1e-10 3e-10 1e-9; do
  timeout 900 python bench.py --steps 20 --warmup 5 --rtol $R >> $OUT/AB_${TAG}.jsonl 2> $OUT/${TAG}_r$R.err
  echo "rtol $R rc=$?"
done
python - <<'PY'
import json
for l in open("gpurun_out/AB_r04a.jsonl"):
    try: d = json.loads(l)
    except Exception: continue
    def w(x):
        return None if not x else (x["value"], x["pcg"]["mean_iterations"], x["guess"]["initial_relres"], (x.get("parity_vs_oracle") or {}).get("mu_zero_mean"), (x.get("parity_vs_oracle") or {}).get("J_n"))
    print(d["config"]["workload"][90:140], "| head", d["value"], d["pcg"]["mean_iterations"], d["pcg"]["guess"], (d.get("parity_vs_oracle") or {}).get("mu_zero_mean"), (d.get("parity_vs_oracle") or {}).get("J_n"), "| vortex", w(d.get("vortex_window")), "| late", w(d.get("late_window")))
PY
