#!/bin/bash
# round 4, batch 15: repairs through the quad routines: bit identity, then the warm-up sweep again
O=gpurun_out/r4t; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -s -k "tile_major or agc_tail" > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
grep -E "passed|failed|rc=|tile-major|^E " $O/tests.log | tail -12
for W in 1024 768 512; do for L in 1056 1568; do
  echo "W=$W L_TM=$L: $(CSDR_AGC_W=$W CSDR_AGC_L_TM=$L STEP_AGC=10 STEP_STEPS=200 timeout 300 python tools/step_time.py 2>&1 | grep -E '^(no timer)' | sed -e 's/; kernel.*//' | tr '\n' ' ')" >> $O/w.txt
done; done
cat $O/w.txt
python tools/agc_bursty_time.py > $O/bursty.txt 2>&1; tail -5 $O/bursty.txt
