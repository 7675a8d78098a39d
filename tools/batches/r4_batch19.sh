#!/bin/bash
O=gpurun_out/r4y; mkdir -p $O
for i in 1 2; do for pad in 0 32 64 1024 4160; do
  if [ $pad = 0 ]; then unset CSDR_EXP_OUT_PAD; else export CSDR_EXP_OUT_PAD=$pad; fi
  echo "pad=$pad: $(STEP_STEPS=600 timeout 300 python tools/step_time.py 2>&1 | grep -E '^(no timer|event pair)' | sed -e 's/ per step.*kernel/ kernel/' | tr '\n' ' ')" >> $O/pad.txt
done; done
cat $O/pad.txt
