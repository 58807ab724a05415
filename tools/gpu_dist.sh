#!/bin/bash
# Decomposed-run session on a 1-GPU box: (a) config 5 cut 8 ways as a DRY RUN (host-callback transport, ranks share the
# GPU: counts and sizes of the exchanges are real, timings are not), with two and with one distributed level;
# (b) one rank forced through the decomposed path over RCCL against the single-GPU path.  bash tools/gpu_dist.sh TAG
TAG=$1
OUT=$PWD/gpurun_out; mkdir -p $OUT
for L in 2 1; do
  timeout 1200 python bench.py --gpus 8 --transport gloo --workload 4M --steps 10 --warmup 5 --preroll 200 --config5 off \
      --no-cpu-baseline --dist-levels $L > $OUT/${TAG}_config5_dry_L$L.json 2> $OUT/${TAG}_config5_dry_L$L.err
  echo "config5 dry run, $L distributed level(s): rc=$?"; python - <<PY
import json
d=json.loads(open("$OUT/${TAG}_config5_dry_L$L.json").read().strip().splitlines()[-1])
print(json.dumps(dict(pcg=d.get("pcg",{}).get("mean_iterations"), comm=d.get("comm_per_step"), levels=d["config"].get("amg_levels"))))
PY
done
for L in 2 1; do
  timeout 600 python bench.py --force-distributed --steps 100 --warmup 20 --no-cpu-baseline --vortex-window off --dist-levels $L \
      > $OUT/${TAG}_1M_forced_L$L.json 2> $OUT/${TAG}_1M_forced_L$L.err
  echo "1M, one rank forced through the decomposed path, $L level(s): rc=$?"; cut -c1-200 $OUT/${TAG}_1M_forced_L$L.json
done
timeout 600 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --vortex-window off > $OUT/${TAG}_1M_single.json 2> $OUT/${TAG}_1M_single.err
echo "1M single: rc=$?"; cut -c1-200 $OUT/${TAG}_1M_single.json
