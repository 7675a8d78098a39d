#!/bin/bash
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "run1024_v3 or second_generation_run_kernels_without" 2>&1 | tail -4
for v in 1 0; do
  echo "== CSDR_RUN1024_V3=$v  FM / DeNo / AGC+FM"
  CSDR_RUN1024_V3=$v STEP_M=1024 STEP_STEPS=100 timeout 300 python tools/step_time.py 2>&1 | grep -E "^region"
  CSDR_RUN1024_V3=$v STEP_M=1024 STEP_DEMOD=none STEP_STEPS=100 timeout 300 python tools/step_time.py 2>&1 | grep -E "^region"
  CSDR_RUN1024_V3=$v STEP_M=1024 STEP_AGC=10 STEP_STEPS=60 timeout 300 python tools/step_time.py 2>&1 | grep -E "^no timer"
done
