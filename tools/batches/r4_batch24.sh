#!/bin/bash
cd /root/repo
for st in 0 1; do
  echo "== stagger $st"
  CSDR_RUN1024_V3_STAGGER=$st STEP_M=1024 STEP_STEPS=200 timeout 300 python tools/step_time.py 2>&1 | grep -E "^region"
done
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "1024" 2>&1 | tail -3
