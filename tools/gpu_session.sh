#!/bin/bash
# Scratch script of the current GPU-box session (overwritten from session to session; the lasting
# recipes are gpu_round.sh, gpu_ab.sh and gpu_kernel_ab.sh).  Run from the repo root via gpurun.
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 2400 python -m pytest tests/test_hip_direct.py -m gpu -q > $OUT/r04s_tests.log 2>&1; echo "tests rc=$?" >> $OUT/r04s_tests.log
tail -4 $OUT/r04s_tests.log; grep -E "^(FAILED|ERROR)|^E  " $OUT/r04s_tests.log | head -20
