#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_api.py -m gpu -q -x 2>&1 | tail -3
bash tools/gpu_ab.sh r04F "TDGL_PCG_PREDICT=last@" "@" "TDGL_PCG_PREDICT=last@" "@" "TDGL_PCG_PREDICT=last@--steps 20 --warmup 5" "@--steps 20 --warmup 5" "TDGL_PCG_PREDICT=last@--workload 250k" "@--workload 250k"
