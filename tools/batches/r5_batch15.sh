#!/bin/bash
# round 5, batch 15: row-major k_agc_spec with the state-only packed quad on whole warm-up blocks: bit identity, then the reference-sized calls
O=gpurun_out/r5p; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "agc" 2>&1 | tail -5 > $O/tests.txt
cat $O/tests.txt
line() { python tools/step_time.py 2>&1 | grep -E '^no timer' | sed 's/no timer *: *//; s/;.*//'; }
for M in 64 256 1024; do
  echo "M=$M nf=4096 fm agc10: $(STEP_M=$M STEP_NF=4096 STEP_DEMOD=fm STEP_AGC=10 STEP_STEPS=400 line)" >> $O/sizes.txt
done
for nf in 1024 2048 8192 16384; do
  echo "M=256 nf=$nf fm agc10: $(STEP_M=256 STEP_NF=$nf STEP_DEMOD=fm STEP_AGC=10 STEP_STEPS=300 line)" >> $O/sizes.txt
done
echo "M=256 nf=4096 none agc10: $(STEP_M=256 STEP_NF=4096 STEP_DEMOD=none STEP_AGC=10 STEP_STEPS=400 line)" >> $O/sizes.txt
cat $O/sizes.txt
