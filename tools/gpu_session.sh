#!/bin/bash
# Scratch script of the current GPU-box session (overwritten from session to session; the lasting
# recipes are gpu_round.sh, gpu_ab.sh and gpu_kernel_ab.sh).  Run from the repo root via gpurun.
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
TAG=r04u
timeout 2400 python -m pytest tests -m gpu -q > $OUT/${TAG}_tests.log 2>&1; echo "tests rc=$?" >> $OUT/${TAG}_tests.log
tail -4 $OUT/${TAG}_tests.log; grep -E "^(FAILED|ERROR)" $OUT/${TAG}_tests.log | head -20
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/prof_${TAG} -o ${TAG} -- python $OLDPWD/bench.py --steps 100 --warmup 20 --no-cpu-baseline --vortex-window off > $OUT/prof_${TAG}_bench.json 2> $OUT/prof_${TAG}_err.log
cd $OLDPWD
DB=$(ls $OUT/prof_${TAG}/*_results.db 2>/dev/null | head -1)
[ -n "$DB" ] && python tools/rocpd_summary.py $DB "rocprofv3 --kernel-trace --stats -- python bench.py --steps 100 --warmup 20 --no-cpu-baseline --vortex-window off" "round 4 ($TAG); MI355X, ROCm 7.2" > $OUT/${TAG}_kernel_stats_1M.txt && grep -E "k_publish_status|k_dot_partial|k_update_xr|k_multi_dot<12>|copyBuffer|total kernel" $OUT/${TAG}_kernel_stats_1M.txt | cut -c1-200
rm -rf $OUT/prof_${TAG}/*.db 2>/dev/null
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/ab_tmp.json 2> $OUT/${TAG}_last.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/ab_tmp.json'))
print("head", d["value"], d["pcg"]["mean_iterations"], "vortex", d["vortex_window"]["value"], "late", d["late_window"]["value"], "sustained", d["sustained"]["value"])
PY
