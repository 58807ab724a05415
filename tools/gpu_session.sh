#!/bin/bash
# Scratch script of the current GPU-box session (overwritten from session to session; the lasting
# recipes are gpu_round.sh, gpu_ab.sh and gpu_kernel_ab.sh).  Run from the repo root via gpurun.
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
TAG=r04c
timeout 1200 python -m pytest tests/test_hip_parity.py -m gpu -q -x > $OUT/${TAG}_tests.log 2>&1; echo "tests rc=$?" >> $OUT/${TAG}_tests.log
tail -5 $OUT/${TAG}_tests.log; grep -E "^(FAILED|ERROR)" $OUT/${TAG}_tests.log | head
: > $OUT/AB_${TAG}.jsonl
for V in 1 0 1 0; do
  if [ $V = 1 ]; then export TDGL_CG_VEC64=1; else unset TDGL_CG_VEC64; fi
  timeout 600 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --rtol 1e-9 --late-steps 0 >> $OUT/AB_${TAG}.jsonl 2> $OUT/${TAG}_v$V.err
  echo "vec64=$V rc=$?"
done
unset TDGL_CG_VEC64
timeout 900 python bench.py --steps 20 --warmup 5 --rtol 1e-9 >> $OUT/AB_${TAG}.jsonl 2> $OUT/${TAG}_full.err
python - <<'PY'
import json
for l in open("gpurun_out/AB_r04c.jsonl"):
    try: d = json.loads(l)
    except Exception: continue
    def w(x):
        return None if not x else (x["value"], x["pcg"]["mean_iterations"], x["guess"]["initial_relres"], (x.get("parity_vs_oracle") or {}).get("mu_zero_mean"), (x.get("parity_vs_oracle") or {}).get("J_n"))
    print(d["config"]["workload"][:30], "| head", d["value"], d["pcg"]["mean_iterations"], d["pcg"]["guess"], (d.get("parity_vs_oracle") or {}).get("mu_zero_mean"), (d.get("parity_vs_oracle") or {}).get("J_n"), "| vortex", w(d.get("vortex_window")), "| late", w(d.get("late_window")))
PY
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/prof_${TAG} -o ${TAG} -- python $OLDPWD/bench.py --steps 100 --warmup 20 --no-cpu-baseline --vortex-window off --rtol 1e-9 > $OUT/prof_${TAG}_bench.json 2> $OUT/prof_${TAG}_err.log
cd $OLDPWD
DB=$(ls $OUT/prof_${TAG}/*_results.db 2>/dev/null | head -1)
[ -n "$DB" ] && python tools/rocpd_summary.py $DB "rocprofv3 --kernel-trace --stats -- python bench.py --steps 100 --warmup 20 --no-cpu-baseline --vortex-window off --rtol 1e-9" "round 4 ($TAG); MI355X, ROCm 7.2" > $OUT/${TAG}_kernel_stats_1M.txt && head -28 $OUT/${TAG}_kernel_stats_1M.txt | cut -c1-200
rm -rf $OUT/prof_${TAG}/*.db 2>/dev/null
exit 0
