#!/bin/bash
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
for W in 5k 60k; do
  cd /tmp
  TDGL_NO_GRAPH=1 timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_r03c_$W -o r03c -- python $OLDPWD/bench.py --workload $W --steps 400 --warmup 20 --no-cpu-baseline --vortex-window off > $OUT/prof_r03c_${W}_bench.json 2> $OUT/prof_r03c_${W}_err.log
  cd $OLDPWD
  DB=$(ls $OUT/prof_r03c_$W/*_results.db | head -1)
  python tools/rocpd_summary.py $DB "rocprofv3 --kernel-trace --stats -- python bench.py --workload $W --steps 400 --warmup 20 --no-cpu-baseline --vortex-window off" "round 3 (r03c); MI355X, ROCm 7.2" > $OUT/r03c_kernel_stats_$W.txt
  head -24 $OUT/r03c_kernel_stats_$W.txt | cut -c1-60,100-190
  python -c "
import json; d=json.load(open('$OUT/prof_r03c_${W}_bench.json')); print('$W', d['value'], d['ms_per_step'], d['pcg']['mean_iterations'])"
  rm -rf $OUT/prof_r03c_$W/*.db
done
