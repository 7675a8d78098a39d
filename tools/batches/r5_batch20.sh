#!/bin/bash
# round 5, batch 20: the fixed no-warm-up test; 4096 channels at the reference chunk with 64-frame runs
O=gpurun_out/r5v; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "without_warm_up or fused4096" 2>&1 | tail -4 > $O/tests.txt
cat $O/tests.txt
line() { python tools/step_time.py 2>&1 | grep -E '^no timer' | sed 's/no timer *: *//; s/;.*//'; }
for nf in 2048 4096 8192; do
  for d in fm none; do
    echo "M=4096 nf=$nf $d: $(STEP_M=4096 STEP_NF=$nf STEP_DEMOD=$d STEP_STEPS=300 line)" >> $O/sizes.txt
  done
done
cat $O/sizes.txt
