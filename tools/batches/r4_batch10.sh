#!/bin/bash
# round 4, batch 10: shortened gain chain: all AGC tests (bit identity, oracle parity), segment-length sweep
O=gpurun_out/r4j; mkdir -p $O
timeout 1700 python -m pytest tests -m gpu -x -q -s -k "agc or tile_major or (run_sized and 10.0) or am_matches or wbfm_matches or pfb1024_fused" > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
grep -E "passed|failed|rc=|tile-major|bit-identical|mismatch|Error|^E |steady" $O/tests.log | tail -40
for L in 0 1568 1824 2080 2336 2592 2848 3104; do
  if [ $L = 0 ]; then unset CSDR_AGC_L_TM; else export CSDR_AGC_L_TM=$L; fi
  echo "L_TM=$L: $(STEP_AGC=10 STEP_STEPS=300 timeout 300 python tools/step_time.py 2>&1 | grep -E '^(no timer)' | sed -e 's/; kernel.*//' | tr '\n' ' ')" >> $O/sweep.txt
done
unset CSDR_AGC_L_TM
echo "row-major: $(CSDR_AGC_TM=0 STEP_AGC=10 STEP_STEPS=300 timeout 300 python tools/step_time.py 2>&1 | grep -E '^(no timer)' | sed -e 's/; kernel.*//' | tr '\n' ' ')" >> $O/sweep.txt
cat $O/sweep.txt
