#!/bin/bash
# Scratch script of the current GPU-box session (overwritten from session to session; the lasting
# recipes are gpu_round.sh, gpu_ab.sh and gpu_kernel_ab.sh).  Run from the repo root via gpurun.
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
TAG=r04d
timeout 2400 python -m pytest tests -m gpu -q > $OUT/${TAG}_tests.log 2>&1; echo "tests rc=$?" >> $OUT/${TAG}_tests.log
tail -5 $OUT/${TAG}_tests.log; grep -E "^(FAILED|ERROR)" $OUT/${TAG}_tests.log | head -20
timeout 600 python tools/diag_guess_trace.py --skip 7500 --steps 160 --rtol 1e-9 --out $OUT/${TAG}_guess_trace_late.json > $OUT/${TAG}_guess_trace_late.txt 2>&1
tail -70 $OUT/${TAG}_guess_trace_late.txt
