#!/bin/bash
# shorter runs for mid-sized calls of k_run1024v3: the 1024-channel tests, and the reference-sized chunk (4096 frames) against k_run1024v2
cd /root/repo
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x -k "1024" 2>&1 | tail -3
for v in 1 0; do
  echo "== CSDR_RUN1024_V3=$v, 4096 / 16384 frames per call"
  CSDR_RUN1024_V3=$v STEP_M=1024 STEP_NF=4096 STEP_STEPS=300 timeout 300 python tools/step_time.py 2>&1 | grep -E "^region"
  CSDR_RUN1024_V3=$v STEP_M=1024 STEP_NF=16384 STEP_STEPS=200 timeout 300 python tools/step_time.py 2>&1 | grep -E "^region"
done
