#!/bin/bash
# Scratch script of the current GPU-box session (overwritten from session to session; the lasting
# recipes are gpu_round.sh, gpu_ab.sh and gpu_kernel_ab.sh).  Run from the repo root via gpurun.
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
TAG=r04t
timeout 2400 python -m pytest tests -m gpu -q > $OUT/${TAG}_tests.log 2>&1; echo "tests rc=$?" >> $OUT/${TAG}_tests.log
tail -4 $OUT/${TAG}_tests.log; grep -E "^(FAILED|ERROR)" $OUT/${TAG}_tests.log | head -20
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/${TAG}_smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/${TAG}_smoke.log | cut -c1-300
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/BENCH_${TAG}_1M_driver.json 2> $OUT/${TAG}_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/BENCH_r04t_1M_driver.json"))
print("head", d["value"], d["pcg"]["mean_iterations"], d["parity_vs_oracle"]["ok"], "vortex", d["vortex_window"]["value"], "late", d["late_window"]["value"], d["late_window"]["parity_vs_oracle"]["ok"], "sustained", d["sustained"]["value"], "roofline", d["roofline"]["frac"], d["roofline"]["traffic_source"][-40:])
PY
