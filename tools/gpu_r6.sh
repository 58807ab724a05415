#!/bin/bash
# Round-6 GPU session: bash tools/gpu_r6.sh TAG STAGE...   (stages: newtests, bench3, fulltests, strip, ...)
TAG=$1; shift
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
mkdir -p $OUT
for STAGE in "$@"; do
case $STAGE in
newtests)
  timeout 1500 python -m pytest tests/test_hip_direct.py -q -x -k "preconditioner or factors_as" > $OUT/${TAG}_t1.log 2>&1; echo "t1 rc=$?"; tail -5 $OUT/${TAG}_t1.log
  timeout 1500 python -m pytest tests/test_hip_fullsize.py -q -k "config4" > $OUT/${TAG}_t2.log 2>&1; echo "t2 rc=$?"; tail -5 $OUT/${TAG}_t2.log
  timeout 900 python -m pytest tests/test_bench_cli.py -q -k "single_gpu or paused" > $OUT/${TAG}_t3.log 2>&1; echo "t3 rc=$?"; tail -5 $OUT/${TAG}_t3.log
  ;;
bench3)
  for M in auto factors vcycle; do
    timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --mu-precond $M > $OUT/BENCH_${TAG}_1M_$M.json 2> $OUT/${TAG}_bench_$M.err
    echo "bench $M rc=$?"; python - <<PY
import json
try:
    d=json.load(open("$OUT/BENCH_${TAG}_1M_$M.json"))
    print({k:d.get(k) for k in ("value",)}, d["pcg"].get("mean_iterations"), d["pcg"].get("preconditioner"), d["setup_s"].get("precond_direct"))
    for w in ("vortex_window","sustained","late_window"):
        x=d.get(w) or {}
        print(w, x.get("value"), (x.get("pcg") or {}).get("mean_iterations"), x.get("preconditioner"), (x.get("guess") or {}))
except Exception as e:
    print("no line", e); print(open("$OUT/${TAG}_bench_$M.err").read()[-1500:])
PY
  done
  ;;
fulltests)
  timeout 2700 python -m pytest tests -m gpu -q > $OUT/${TAG}_tests.log 2>&1; echo "tests rc=$?"; tail -6 $OUT/${TAG}_tests.log; grep -E "^(FAILED|ERROR)" $OUT/${TAG}_tests.log | head -20
  ;;
switchdiag)
  timeout 600 python tools/diag_switch_oracle.py > $OUT/${TAG}_switchdiag.log 2>&1; echo "switchdiag rc=$?"; tail -30 $OUT/${TAG}_switchdiag.log
  ;;
prof)
  # kernel trace of the bench with $PROF_ARGS (default: the factors forced) -> ${TAG}_kernel_stats_<name>.txt
  PA=${PROF_ARGS:---mu-precond factors}
  PN=${PROF_NAME:-1M_factors}
  cd /tmp
  timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/prof_${TAG} -o ${TAG} -- python $OLDPWD/bench.py --steps 100 --warmup 20 --no-cpu-baseline --vortex-window off $PA > $OUT/prof_${TAG}_bench.json 2> $OUT/prof_${TAG}_err.log
  cd $OLDPWD
  DB=$(ls $OUT/prof_${TAG}/*_results.db 2>/dev/null | head -1)
  [ -n "$DB" ] && python tools/rocpd_summary.py $DB "rocprofv3 --kernel-trace --stats -- python bench.py --steps 100 --warmup 20 --no-cpu-baseline --vortex-window off $PA" "round 6 ($TAG); MI355X, ROCm 7.2" > $OUT/${TAG}_kernel_stats_${PN}.txt && head -40 $OUT/${TAG}_kernel_stats_${PN}.txt | cut -c1-220
  rm -rf $OUT/prof_${TAG}/*.db 2>/dev/null
  ;;
pdmid)
  # mid sizes: the product default (fp64 direct solve in the run-ahead loop + the loop's solver choice) against the
  # two-preconditioner CG (fp32 factors / V-cycle per solve) forced down to 100k sites
  for W in ${PD_WORKLOADS:-250k 450k strip250k strip500k}; do
    for V in default pd; do
      EXTRA=""; [ $V = pd ] && EXTRA="--sub-limits 32000,100000"
      timeout 900 python bench.py --workload $W --steps 20 --warmup 5 --no-cpu-baseline --late-steps 3000 $EXTRA > $OUT/BENCH_${TAG}_${W}_$V.json 2> $OUT/${TAG}_${W}_$V.err
      python - <<PY
import json
try:
    d=json.load(open("$OUT/BENCH_${TAG}_${W}_$V.json"))
    pr=lambda x: (x.get("value"), (x.get("pcg") or {}).get("mean_iterations"), (x.get("preconditioner") or x.get("pcg",{}).get("preconditioner") or {}).get("solves_factors"), x.get("direct_switching") or x.get("pcg",{}).get("direct_switching"))
    print("$W $V", "headline", pr(d), "| vortex", pr(d.get("vortex_window") or {}), "| sustained", (d.get("sustained") or {}).get("value"), "| late", pr(d.get("late_window") or {}), "| setup", d["setup_s"]["total"])
except Exception as e:
    print("$W $V no line", e); print(open("$OUT/${TAG}_${W}_$V.err").read()[-800:])
PY
    done
  done
  ;;
symab)
  # symmetric tiles on the first level (k_sub_down_sym) against whole blocks (TDGL_PD_SYM=0): factors forced, 1M
  for V in 1 0; do
    TDGL_PD_SYM=$V timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --mu-precond factors --late-steps 2000 > $OUT/BENCH_${TAG}_1M_sym$V.json 2> $OUT/${TAG}_sym$V.err
    python - <<PY
import json
try:
    d=json.load(open("$OUT/BENCH_${TAG}_1M_sym$V.json"))
    print("sym=$V", d["value"], d.get("roofline_precond"), "late", (d.get("late_window") or {}).get("value"), "sustained", (d.get("sustained") or {}).get("value"), "parity", (d.get("parity") or {}).get("max_err"), d["setup_s"].get("precond_direct"))
except Exception as e:
    print("sym=$V no line", e); print(open("$OUT/${TAG}_sym$V.err").read()[-1500:])
PY
  done
  ;;
*) echo "unknown stage $STAGE";;
esac
done
exit 0
