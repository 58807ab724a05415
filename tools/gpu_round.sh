#!/bin/bash
# One GPU-box session: tests, bench line, kernel trace, PMC passes.  Usage (from the repo root, via gpurun):
#   bash tools/gpu_round.sh TAG [tests|notests] [pmc|nopmc] [extra bench args...]
# Outputs land in gpurun_out/ (scratch); summaries worth keeping are copied to profiles/ by hand.
TAG=$1; TESTS=${2:-tests}; PMC=${3:-nopmc}; shift 3
ROUND=${TAG:0:3}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
mkdir -p $OUT
if [ "$TESTS" = tests ]; then
  timeout 2700 python -m pytest tests -m gpu -q > $OUT/${TAG}_tests.log 2>&1
  echo "tests rc=$?" >> $OUT/${TAG}_tests.log
  tail -4 $OUT/${TAG}_tests.log; grep -E "^(FAILED|ERROR)" $OUT/${TAG}_tests.log | head
fi
timeout 900 python bench.py --steps 20 --warmup 5 --trace-iterations $OUT/${TAG}_trace_1M.json "$@" > $OUT/BENCH_${TAG}_1M_driver.json 2> $OUT/${TAG}_bench_driver.err
echo "bench(driver flags) rc=$?"; cat $OUT/BENCH_${TAG}_1M_driver.json | cut -c1-300
timeout 900 python bench.py --no-cpu-baseline "$@" > $OUT/BENCH_${TAG}_1M.json 2> $OUT/${TAG}_bench.err
echo "bench(default) rc=$?"; cat $OUT/BENCH_${TAG}_1M.json | cut -c1-300
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/prof_${TAG} -o ${TAG} -- python $OLDPWD/bench.py --steps 100 --warmup 20 --no-cpu-baseline --vortex-window off "$@" > $OUT/prof_${TAG}_bench.json 2> $OUT/prof_${TAG}_err.log
cd $OLDPWD
DB=$(ls $OUT/prof_${TAG}/*_results.db 2>/dev/null | head -1)
[ -n "$DB" ] && python tools/rocpd_summary.py $DB "rocprofv3 --kernel-trace --stats -- python bench.py --steps 100 --warmup 20 --no-cpu-baseline --vortex-window off $*" "round ${ROUND:1} ($TAG); MI355X, ROCm 7.2" > $OUT/${TAG}_kernel_stats_1M.txt && head -32 $OUT/${TAG}_kernel_stats_1M.txt | cut -c1-200
if [ "$PMC" = pmc ]; then
  cd /tmp
  timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_${TAG}_f -o f -- python $OLDPWD/bench.py --steps 20 --warmup 5 --preroll 100 --no-cpu-baseline --vortex-window off "$@" > /dev/null 2> $OUT/pmc_${TAG}_f.err
  timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_${TAG}_w -o w -- python $OLDPWD/bench.py --steps 20 --warmup 5 --preroll 100 --no-cpu-baseline --vortex-window off "$@" > /dev/null 2> $OUT/pmc_${TAG}_w.err
  cd $OLDPWD
  F=$(ls $OUT/pmc_${TAG}_f/*_results.db | head -1); W=$(ls $OUT/pmc_${TAG}_w/*_results.db | head -1)
  python tools/rocpd_pmc.py $F $W --json $OUT/${TAG}_pmc_hbm_traffic_1M.json "HBM traffic per launch from rocprofv3 PMC counters, 1M-site workload, MI355X, ROCm 7.2, round ${ROUND:1} ($TAG)" "two separate passes: rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE -- python bench.py --steps 20 --warmup 5 --preroll 100 --no-cpu-baseline --vortex-window off" > $OUT/${TAG}_pmc_hbm_traffic_1M.txt
  head -12 $OUT/${TAG}_pmc_hbm_traffic_1M.txt | cut -c1-170
  # the databases are large: keep only the summaries
  rm -rf $OUT/pmc_${TAG}_f $OUT/pmc_${TAG}_w
fi
rm -rf $OUT/prof_${TAG}/*.db 2>/dev/null
exit 0
