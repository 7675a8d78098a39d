#!/bin/bash
# round 5, batch 27: k_run1024v3 without warm-up windows (nowu + k_run1024_dcfix): parity, then A/B against CSDR_NOWU=0
O=gpurun_out/r6h; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -s -k "1024" 2>&1 | grep -E "no-warm-up|passed|failed|Error|assert" | head -30 > $O/tests.txt
cat $O/tests.txt
for i in 1 2; do
  for nw in 1 0; do
    echo "fm CSDR_NOWU=$nw: $(CSDR_NOWU=$nw STEP_M=1024 STEP_DEMOD=fm STEP_STEPS=400 python tools/step_time.py 2>&1 | grep -E '^event pair|^no timer' | sed 's/ per step.*kernel/ kernel/; s/ per step.*//' | tr '\n' ' ')" >> $O/ab.txt
  done
done
for nw in 1 0; do
  echo "none CSDR_NOWU=$nw: $(CSDR_NOWU=$nw STEP_M=1024 STEP_DEMOD=none STEP_STEPS=400 python tools/step_time.py 2>&1 | grep -E '^event pair|^no timer' | sed 's/ per step.*kernel/ kernel/; s/ per step.*//' | tr '\n' ' ')" >> $O/ab.txt
done
cat $O/ab.txt
