#!/bin/bash
# k_run1024v3 (one workgroup per CU, output lines staged in registers) against k_run1024v2: parity, then step / kernel time
cd /root/repo
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -s -k "run1024_v3" 2>&1 | tail -8
for v in 0 1; do
  echo "== CSDR_RUN1024_V3=$v"
  CSDR_RUN1024_V3=$v STEP_M=1024 STEP_STEPS=60 timeout 300 python tools/step_time.py 2>&1 | tail -3
done
