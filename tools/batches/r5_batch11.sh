#!/bin/bash
# round 5, batch 11: the reference's own chunk size (4096 frames per call) per M and tail, and what a 4096-frame AGC call is made of
O=gpurun_out/r5k; mkdir -p $O
line() { python tools/step_time.py 2>&1 | grep -E '^no timer' | sed 's/no timer *: *//; s/;.*//'; }
for M in 64 256 1024 4096; do
  for d in fm none; do
    echo "M=$M nf=4096 $d: $(STEP_M=$M STEP_NF=4096 STEP_DEMOD=$d STEP_STEPS=400 line)" >> $O/sizes.txt
  done
  echo "M=$M nf=4096 fm agc10: $(STEP_M=$M STEP_NF=4096 STEP_DEMOD=fm STEP_AGC=10 STEP_STEPS=400 line)" >> $O/sizes.txt
done
for nf in 1024 2048 8192 16384 65536; do
  echo "M=256 nf=$nf fm agc10: $(STEP_M=256 STEP_NF=$nf STEP_DEMOD=fm STEP_AGC=10 STEP_STEPS=300 line)" >> $O/sizes.txt
done
cat $O/sizes.txt
export TMPDIR=/tmp
export STEP_M=256 STEP_NF=4096 STEP_DEMOD=fm STEP_AGC=10 STEP_STEPS=300
rocprofv3 --kernel-trace --stats -d /tmp/prof_agc4096 -o agc4096 -- python3 tools/step_time.py > $O/prof_agc4096.log 2>&1
f=$(find /tmp/prof_agc4096 -name '*kernel_stats.csv' | head -1); cp "$f" $O/agc4096_kernel_stats.csv; head -12 $O/agc4096_kernel_stats.csv | cut -c1-200
du -sh gpurun_out
