#!/bin/bash
# k_run1024v3 with runs of one block for short calls: the 1024-channel tests, then small-call times
cd /root/repo
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x -k "1024" 2>&1 | tail -3
for d in fm none; do for nf in 256 1024 4096 16384; do
  echo -n "$d nf=$nf: "; STEP_M=1024 STEP_NF=$nf STEP_DEMOD=$d STEP_STEPS=300 timeout 300 python tools/step_time.py 2>&1 | grep -E "^region" | cut -c27-
done; done
