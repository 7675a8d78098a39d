#!/bin/bash
# round 4, batch 5: k_agc_spec_tm with gain / post wave roles: bit identity, then the segment-length sweep
O=gpurun_out/r4e; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q -s -k "tile_major or agc_tail_full_size or agc_tail_steady or (run_sized and 10.0)" > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
grep -E "passed|failed|rc=|tile-major|bit-identical|mismatch|Error|error|assert|steady" $O/tests.log | tail -30
for L in 1040 1280 1536 1792 2048; do
  echo "L_TM=$L: $(CSDR_AGC_L_TM=$L STEP_AGC=10 STEP_STEPS=300 timeout 300 python tools/step_time.py 2>&1 | grep -E '^(no timer)' | sed -e 's/; kernel.*//' | tr '\n' ' ')" >> $O/sweep.txt
done
echo "row-major: $(CSDR_AGC_TM=0 STEP_AGC=10 STEP_STEPS=300 timeout 300 python tools/step_time.py 2>&1 | grep -E '^(no timer)' | sed -e 's/; kernel.*//' | tr '\n' ' ')" >> $O/sweep.txt
cat $O/sweep.txt
