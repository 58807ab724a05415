#!/bin/bash
# Scratch script of the current GPU-box session (overwritten from session to session; the lasting
# recipes are gpu_round.sh, gpu_ab.sh and gpu_kernel_ab.sh).  Run from the repo root via gpurun.
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
TAG=r04f
timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_hip_direct.py tests/test_hip_distributed.py -m gpu -q -k "guess or guard or 59k or projection or multi_rank or gram" > $OUT/${TAG}_tests.log 2>&1; echo "tests rc=$?" >> $OUT/${TAG}_tests.log
tail -5 $OUT/${TAG}_tests.log; grep -E "^(FAILED|ERROR)" $OUT/${TAG}_tests.log | head -20
: > $OUT/AB_${TAG}.jsonl
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --force-distributed --vortex-window off >> $OUT/AB_${TAG}.jsonl 2> $OUT/${TAG}_dist1.err
echo "dist1 rc=$?"
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --vortex-window off >> $OUT/AB_${TAG}.jsonl 2> $OUT/${TAG}_single.err
echo "single rc=$?"
timeout 900 python bench.py --gpus 4 --transport gloo --steps 5 --warmup 2 --preroll 20 --no-cpu-baseline --config5 off --timeout 800 >> $OUT/AB_${TAG}.jsonl 2> $OUT/${TAG}_dry4.err
echo "dry4 rc=$?"; tail -3 $OUT/${TAG}_dry4.err
python - <<'PY'
import json
for l in open("gpurun_out/AB_r04f.jsonl"):
    try: d = json.loads(l)
    except Exception: continue
    print(d["config"]["parallelism"][:60], "| head", d["value"], d["pcg"]["mean_iterations"], d["pcg"]["guess"], d.get("host"), d.get("comm_per_step"))
PY
