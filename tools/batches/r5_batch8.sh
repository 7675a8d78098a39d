#!/bin/bash
# round 5, batch 8: AGC tail on L2-resident calls takes every lane (Lmin 16 instead of 384): bit identity + per-call times at the reference chunk sizes
O=gpurun_out/r5h; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q -k "agc" > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
tail -4 $O/tests.log
for nf in 4096 8192 16384 65536; do
  echo "M=256 FM + AGC nf=$nf: $(STEP_NF=$nf STEP_AGC=10 STEP_STEPS=200 python tools/step_time.py 2>&1 | grep -E '^no timer' | sed 's/;.*//')" >> $O/agc_small.txt
done
for nf in 4096; do
  echo "M=1024 FM + AGC nf=$nf: $(STEP_M=1024 STEP_NF=$nf STEP_AGC=10 STEP_STEPS=200 python tools/step_time.py 2>&1 | grep -E '^no timer' | sed 's/;.*//')" >> $O/agc_small.txt
  echo "M=64 FM + AGC nf=$nf: $(STEP_M=64 STEP_NF=$nf STEP_AGC=10 STEP_STEPS=200 python tools/step_time.py 2>&1 | grep -E '^no timer' | sed 's/;.*//')" >> $O/agc_small.txt
done
cat $O/agc_small.txt
