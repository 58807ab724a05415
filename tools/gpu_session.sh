#!/bin/bash
# Scratch script of the current GPU-box session (overwritten from session to session; the lasting
# recipes are gpu_round.sh, gpu_ab.sh and gpu_kernel_ab.sh).  Run from the repo root via gpurun.
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
TAG=r04p
timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_hip_distributed.py -m gpu -q -k "guess or projection or window or multi_rank or gram or more_ranks" > $OUT/${TAG}_tests.log 2>&1; echo "tests rc=$?" >> $OUT/${TAG}_tests.log
tail -4 $OUT/${TAG}_tests.log; grep -E "^(FAILED|ERROR)|^E  " $OUT/${TAG}_tests.log | head -20
: > $OUT/AB_${TAG}.jsonl
for W in 0 12 0 12; do
  timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --guess-window $W > $OUT/ab_tmp.json 2> $OUT/${TAG}_last.err
  python - "$W" <<'PY' >> $OUT/AB_r04p.jsonl
import json,sys
d=json.load(open('gpurun_out/ab_tmp.json'))
print(json.dumps(dict(window=sys.argv[1], head=d["value"], its=d["pcg"]["mean_iterations"], guess=d["pcg"]["guess"], vortex=d["vortex_window"]["value"], vortex_its=d["vortex_window"]["pcg"]["mean_iterations"], sustained=d["sustained"]["value"], sustained_its=d["sustained"]["pcg"]["mean_iterations"], late=d["late_window"]["value"], late_its=d["late_window"]["pcg"]["mean_iterations"], late_guess=d["late_window"]["guess"])))
PY
  tail -1 $OUT/AB_r04p.jsonl
done
echo "{\"soak\": \"1M 30000 auto window\"}" > $OUT/SOAK_r04p.jsonl
timeout 900 python tools/soak.py 1M 30000 >> $OUT/SOAK_r04p.jsonl 2> $OUT/soak_last.err; echo "soak rc=$?"; tail -2 $OUT/SOAK_r04p.jsonl | cut -c1-300
