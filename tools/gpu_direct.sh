#!/bin/bash
# Direct-solve session on a GPU box: the direct-solve tests, bench lines at the sizes the direct mu solves cover, and a kernel
# trace of the 250k-site workload.  bash tools/gpu_direct.sh TAG
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
TAG=$1
timeout 1500 python -m pytest tests/test_hip_direct.py -m gpu -x -q 2>&1 | tail -4
for W in 250k 120k 60k 23k; do
timeout 900 python bench.py --workload $W --steps 200 --warmup 20 --no-cpu-baseline --vortex-window off > $OUT/${TAG}_$W.json 2> $OUT/${TAG}_$W.err
python - <<PY
import json
d=json.loads(open("$OUT/${TAG}_$W.json").read().strip().splitlines()[-1])
print("$W", d["value"], d.get("roofline_direct",{}).get("avg_solve_ms"), d.get("roofline_direct",{}).get("achieved"))
PY
done
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/prof_$TAG -o $TAG -- python $OLDPWD/bench.py --workload 250k --steps 100 --warmup 20 --no-cpu-baseline --vortex-window off > $OUT/prof_${TAG}_bench.json 2> $OUT/prof_${TAG}_err.log
cd $OLDPWD
DB=$(ls $OUT/prof_$TAG/*_results.db | head -1)
python tools/rocpd_summary.py $DB "250k two-level direct" "$TAG" > $OUT/${TAG}_kernel_stats_250k.txt; head -12 $OUT/${TAG}_kernel_stats_250k.txt | cut -c1-60,108-150
rm -rf $OUT/prof_$TAG/*.db
