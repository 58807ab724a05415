#!/bin/bash
mkdir -p gpurun_out
python tools/diag_setup.py 1000 --profile > gpurun_out/setup_profile_1M_B.txt 2>&1
head -48 gpurun_out/setup_profile_1M_B.txt
