#!/bin/bash
# Same-box A/B at kernel level: bash tools/gpu_kernel_ab.sh TAG "ENV=.. @ bench args" ...
# For every variant: rocprofv3 --kernel-trace of bench.py (100 timed steps) and the average duration of
# every kernel with > 0.5 % of the time -> gpurun_out/KAB_<TAG>.txt
TAG=$1; shift
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
: > $OUT/KAB_${TAG}.txt
I=0
for V in "$@"; do
  I=$((I+1))
  E=""; A="$V"
  case "$V" in *@*) E="${V%%@*}"; A="${V#*@}";; esac
  cd /tmp
  timeout 900 env $E rocprofv3 --kernel-trace --stats -d $OUT/kab_${TAG}_$I -o k -- python $OLDPWD/bench.py --steps 100 --warmup 20 --no-cpu-baseline --vortex-window off $A > $OUT/kab_${TAG}_$I.json 2> $OUT/kab_${TAG}_$I.err
  cd $OLDPWD
  DB=$(ls $OUT/kab_${TAG}_$I/*_results.db 2>/dev/null | head -1)
  echo "== variant $I: $V" >> $OUT/KAB_${TAG}.txt
  python -c "
import json; d=json.load(open('$OUT/kab_${TAG}_$I.json')); print('   bench (under the profiler):', d['value'], 'steps/s', d['ms_per_step'], 'ms', d['pcg']['mean_iterations'], 'it')" >> $OUT/KAB_${TAG}.txt 2>&1
  [ -n "$DB" ] && python tools/rocpd_summary.py $DB "$V" "kernel A/B $TAG" | awk 'NR>3 && $NF+0 > 0.5 {printf "   %-70s calls %6s avg_ns %10s  %5s%%\n", substr($0,1,70), $(NF-5), $(NF-3), $NF}' >> $OUT/KAB_${TAG}.txt
  rm -rf $OUT/kab_${TAG}_$I
done
cat $OUT/KAB_${TAG}.txt
