#!/bin/bash
# Scratch script of the current GPU-box session (overwritten from session to session; the lasting
# recipes are gpu_round.sh, gpu_ab.sh and gpu_kernel_ab.sh).  Run from the repo root via gpurun.
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
TAG=r04g
: > $OUT/AB_${TAG}.jsonl
for W in 8 10 12 16; do
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --guess-window $W >> $OUT/AB_${TAG}.jsonl 2> $OUT/${TAG}_w$W.err
  echo "window $W rc=$?"
done
for CUT in 1e-18 1e-21 1e-28; do
  TDGL_GUESS_CUT=$CUT timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline >> $OUT/AB_${TAG}.jsonl 2> $OUT/${TAG}_c$CUT.err
  echo "cut $CUT rc=$?"
done
python - <<'PY'
import json
for l in open("gpurun_out/AB_r04g.jsonl"):
    try: d = json.loads(l)
    except Exception: continue
    def w(x):
        return None if not x else (x["value"], x["pcg"]["mean_iterations"], (x.get("guess") or {}).get("initial_relres"))
    print("head", d["value"], d["pcg"]["mean_iterations"], d["pcg"]["guess"], "| vortex", w(d.get("vortex_window")), "| late", w(d.get("late_window")), "| sustained", w(d.get("sustained")))
PY
