#!/bin/bash
# round 5, batch 25: k_run64v2 without warm-up windows (nowu + k_run64_dcfix): parity, then A/B against CSDR_NOWU=0
O=gpurun_out/r6f; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -s -k "run64 or cfg2 or dc_block_off or without_warm_up" 2>&1 | grep -E "no-warm-up|run64v2:|passed|failed|Error|assert" | head -20 > $O/tests.txt
cat $O/tests.txt
for i in 1 2; do
  for nw in 1 0; do
    echo "CSDR_NOWU=$nw: $(CSDR_NOWU=$nw STEP_M=64 STEP_DEMOD=none STEP_STEPS=400 python tools/step_time.py 2>&1 | grep -E '^event pair|^no timer' | sed 's/ per step.*kernel/ kernel/; s/ per step.*//' | tr '\n' ' ')" >> $O/ab.txt
  done
done
cat $O/ab.txt
