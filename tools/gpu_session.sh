#!/bin/bash
# Scratch script of the current GPU-box session (overwritten from session to session; the lasting
# recipes are gpu_round.sh, gpu_ab.sh and gpu_kernel_ab.sh).  Run from the repo root via gpurun.
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
TAG=r04l
timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_hip_direct.py tests/test_hip_distributed.py -m gpu -q -k "window_sizes or error_path or more_ranks or run_ahead_loop or guard" > $OUT/${TAG}_tests.log 2>&1; echo "tests rc=$?" >> $OUT/${TAG}_tests.log
tail -5 $OUT/${TAG}_tests.log; grep -E "^(FAILED|ERROR)|^E  " $OUT/${TAG}_tests.log | head -30
