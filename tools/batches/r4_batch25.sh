#!/bin/bash
cd /root/repo
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "run1024_v3 or second_generation_run_kernels_without" 2>&1 | tail -3
python tools/trace_run1024v3.py 2>&1 | tail -12
STEP_M=1024 STEP_STEPS=200 timeout 300 python tools/step_time.py 2>&1 | grep -E "^region"
