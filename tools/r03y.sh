#!/bin/bash
# closing GPU session of round 3: the 1M evidence again with the final kernels (bench lines, kernel trace, PMC), PMC traffic of
# the direct-solve kernels at 5.8k and 59k sites, bench lines with the CPU leg for the direct paths
bash tools/gpu_round.sh r03y notests pmc
OUT=$PWD/gpurun_out
export TMPDIR=/tmp
for W in 5k 60k; do
  cd /tmp
  timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_r03y_${W}_f -o f -- python $OLDPWD/bench.py --workload $W --steps 200 --warmup 20 --no-cpu-baseline --vortex-window off > /dev/null 2> $OUT/pmc_r03y_${W}_f.err
  timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_r03y_${W}_w -o w -- python $OLDPWD/bench.py --workload $W --steps 200 --warmup 20 --no-cpu-baseline --vortex-window off > /dev/null 2> $OUT/pmc_r03y_${W}_w.err
  cd $OLDPWD
  F=$(ls $OUT/pmc_r03y_${W}_f/*_results.db | head -1); Wd=$(ls $OUT/pmc_r03y_${W}_w/*_results.db | head -1)
  python tools/rocpd_pmc.py $F $Wd "HBM traffic per launch from rocprofv3 PMC counters, workload $W (direct mu solve), MI355X, ROCm 7.2, round 3 (r03y)" "two separate passes: rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE -- python bench.py --workload $W --steps 200 --warmup 20 --no-cpu-baseline --vortex-window off" > $OUT/r03y_pmc_hbm_traffic_${W}.txt
  head -14 $OUT/r03y_pmc_hbm_traffic_${W}.txt | cut -c1-170
  rm -rf $OUT/pmc_r03y_${W}_f $OUT/pmc_r03y_${W}_w
done
: > $OUT/BENCH_r03y_parity_lines.jsonl
for W in 9k 60k; do
  timeout 900 python bench.py --workload $W --steps 20 --warmup 5 > $OUT/tmp_line.json 2> $OUT/r03y_parity.err
  echo "$W rc=$?"; grep "parity" $OUT/r03y_parity.err | cut -c1-400
  cat $OUT/tmp_line.json >> $OUT/BENCH_r03y_parity_lines.jsonl
done
exit 0
