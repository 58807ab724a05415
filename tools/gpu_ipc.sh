#!/bin/bash
# Peer-mapped transport with the ranks SHARING one GPU (a dry run: what is exercised is the mechanism -- IPC inboxes,
# in-kernel flags, the self-test, the automatic choice of transport -- and the launch overhead it adds; the GPU's time
# is divided among the ranks, so steps/s say nothing about scaling).  bash tools/gpu_ipc.sh TAG
TAG=$1
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 600 python -m pytest tests/test_hip_distributed.py -m gpu -x -q -k "selftest" 2>&1 | tail -15
timeout 600 python bench.py --gpus 2 --share-devices --steps 50 --warmup 10 --no-cpu-baseline --config5 off > $OUT/${TAG}_1M_2ranks_auto.json 2> $OUT/${TAG}_1M_2ranks_auto.err
echo "1M, 2 ranks sharing the GPU, transport auto: rc=$?"; python - <<PY
import json
d=json.loads(open("$OUT/${TAG}_1M_2ranks_auto.json").read().strip().splitlines()[-1])
print(json.dumps(dict(value=d.get("value"), pcg=d.get("pcg",{}).get("mean_iterations"), transport=d.get("transport"), comm=d.get("comm_per_step")))[:3000])
PY
timeout 1200 python bench.py --gpus 8 --share-devices --transport ipc --workload 4M --steps 10 --warmup 5 --preroll 200 --config5 off --no-cpu-baseline > $OUT/${TAG}_config5_ipc_shared.json 2> $OUT/${TAG}_config5_ipc_shared.err
echo "config 5, 8 ranks sharing the GPU, peer-mapped transport: rc=$?"; python - <<PY
import json
d=json.loads(open("$OUT/${TAG}_config5_ipc_shared.json").read().strip().splitlines()[-1])
print(json.dumps(dict(value=d.get("value"), pcg=d.get("pcg",{}).get("mean_iterations"), transport=d.get("transport"), comm=d.get("comm_per_step")))[:3000])
PY
tail -3 $OUT/${TAG}_config5_ipc_shared.err
