#!/bin/bash
# round 4, batch 14: shorter AGC warm-up now that repairs stop at checkpoints
O=gpurun_out/r4r; mkdir -p $O
for W in 1024 768 512 384; do for L in 1056 1568 2080; do
  echo "W=$W L_TM=$L: $(CSDR_AGC_W=$W CSDR_AGC_L_TM=$L STEP_AGC=10 STEP_STEPS=200 timeout 300 python tools/step_time.py 2>&1 | grep -E '^(no timer)' | sed -e 's/; kernel.*//' | tr '\n' ' ')" >> $O/w.txt
done; done
cat $O/w.txt
