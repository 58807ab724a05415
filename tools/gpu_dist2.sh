#!/bin/bash
TAG=$1
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_hip_distributed.py tests/test_hip_distributed_fullsize.py -m gpu -q 2>&1 | tail -8
timeout 600 python bench.py --gpus 2 --share-devices --steps 30 --warmup 5 --preroll 60 --no-cpu-baseline --config5 off > $OUT/${TAG}_1M_2ranks_auto.json 2> $OUT/${TAG}_1M_2ranks_auto.err
echo "1M, 2 ranks sharing the GPU, transport auto: rc=$?"; python - <<PY
import json
d=json.loads(open("$OUT/${TAG}_1M_2ranks_auto.json").read().strip().splitlines()[-1])
print(json.dumps(dict(value=d.get("value"), pcg=d.get("pcg",{}).get("mean_iterations"), transport=d.get("transport"), par=d["config"]["parallelism"]))[:3000])
PY
tail -5 $OUT/${TAG}_1M_2ranks_auto.err
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/prof_${TAG} -o ${TAG} -- python $OLDPWD/bench.py --force-distributed --steps 100 --warmup 20 --no-cpu-baseline --vortex-window off > $OUT/prof_${TAG}_bench.json 2> $OUT/prof_${TAG}_err.log
cd $OLDPWD
DB=$(ls $OUT/prof_${TAG}/*_results.db 2>/dev/null | head -1)
[ -n "$DB" ] && python tools/rocpd_summary.py $DB "rocprofv3 --kernel-trace --stats -- python bench.py --force-distributed --steps 100 --warmup 20 --no-cpu-baseline --vortex-window off" "round 5 ($TAG): ONE rank forced through the decomposed path (two distributed levels, RCCL world 1); MI355X, ROCm 7.2" > $OUT/${TAG}_kernel_stats_1M_forced_distributed.txt && head -30 $OUT/${TAG}_kernel_stats_1M_forced_distributed.txt | cut -c1-200
rm -rf $OUT/prof_${TAG}/*.db
