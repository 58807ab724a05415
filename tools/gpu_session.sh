#!/bin/bash
# Scratch script of the current GPU-box session (overwritten from session to session; the lasting
# recipes are gpu_round.sh, gpu_ab.sh and gpu_kernel_ab.sh).  Run from the repo root via gpurun.
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
TAG=r04k
: > $OUT/AB_${TAG}.jsonl
timeout 900 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --workload 4M --late-steps 1000 >> $OUT/AB_${TAG}.jsonl 2> $OUT/${TAG}_4M.err
echo "4M rc=$?"
timeout 1500 python bench.py --gpus 8 --transport gloo --workload 4M --steps 5 --warmup 2 --preroll 20 --no-cpu-baseline --config5 off --timeout 1400 >> $OUT/AB_${TAG}.jsonl 2> $OUT/${TAG}_dry8.err
echo "dry8 rc=$?"; tail -3 $OUT/${TAG}_dry8.err
for WL in 5k 23k; do
  timeout 600 python bench.py --steps 2000 --warmup 200 --no-cpu-baseline --workload $WL --late-steps 0 >> $OUT/AB_${TAG}.jsonl 2> $OUT/${TAG}_$WL.err
  echo "$WL rc=$?"
done
python - <<'PY'
import json
for l in open("gpurun_out/AB_r04k.jsonl"):
    try: d = json.loads(l)
    except Exception: continue
    def w(x):
        return None if not x else (x["value"], x["pcg"]["mean_iterations"])
    print(d["config"]["workload"][:34], d["config"]["parallelism"][:40], "| head", d["value"], d["pcg"]["mean_iterations"], "| vortex", w(d.get("vortex_window")), "| sustained", w(d.get("sustained")), "| late", w(d.get("late_window")), d.get("comm_per_step"), d["setup_s"].get("total"))
PY
