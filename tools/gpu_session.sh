#!/bin/bash
# Scratch script of the current GPU-box session (overwritten from session to session; the lasting
# recipes are gpu_round.sh, gpu_ab.sh and gpu_kernel_ab.sh).  Run from the repo root via gpurun.
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
TAG=r04e
timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_hip_direct.py tests/test_hip_distributed.py -m gpu -q -k "guess or guard or 59k or projection or multi_rank or gram" > $OUT/${TAG}_tests.log 2>&1; echo "tests rc=$?" >> $OUT/${TAG}_tests.log
tail -5 $OUT/${TAG}_tests.log; grep -E "^(FAILED|ERROR)" $OUT/${TAG}_tests.log | head -20
: > $OUT/AB_${TAG}.jsonl
timeout 900 python bench.py --steps 20 --warmup 5 >> $OUT/AB_${TAG}.jsonl 2> $OUT/${TAG}_full.err
echo "full rc=$?"
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --force-distributed --vortex-window off >> $OUT/AB_${TAG}.jsonl 2> $OUT/${TAG}_dist1.err
echo "dist1 rc=$?"
python - <<'PY'
import json
for l in open("gpurun_out/AB_r04e.jsonl"):
    try: d = json.loads(l)
    except Exception: continue
    def w(x):
        return None if not x else (x["value"], x["pcg"]["mean_iterations"], (x.get("guess") or {}).get("initial_relres"), (x.get("parity_vs_oracle") or {}).get("mu_zero_mean"), (x.get("parity_vs_oracle") or {}).get("J_n"))
    print(d["config"]["parallelism"][:20], "| head", d["value"], d["pcg"]["mean_iterations"], d["pcg"]["guess"], (d.get("parity_vs_oracle") or {}).get("mu_zero_mean"), (d.get("parity_vs_oracle") or {}).get("J_n"), "| vortex", w(d.get("vortex_window")), "| late", w(d.get("late_window")), "| sustained", w(d.get("sustained")), d.get("host"))
PY
