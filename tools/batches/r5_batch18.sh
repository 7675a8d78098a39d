#!/bin/bash
# round 5, batch 18: tile-major AGC tail on mid-sized calls: does a shorter minimum segment (more workgroups, more re-reads out of the L2 / MALL) pay?
O=gpurun_out/r5s; mkdir -p $O
line() { python tools/step_time.py 2>&1 | grep -E '^no timer' | sed 's/no timer *: *//; s/;.*//'; }
for cfg in "256 16384" "256 32768" "256 65536" "1024 4096" "1024 16384" "64 65536"; do
  set -- $cfg
  for lm in 384 256 192 128 64; do
    echo "M=$1 nf=$2 fm agc10 Lmin_tm=$lm: $(CSDR_AGC_LMIN_TM=$lm STEP_M=$1 STEP_NF=$2 STEP_DEMOD=fm STEP_AGC=10 STEP_STEPS=200 line)" >> $O/lmin.txt
  done
done
cat $O/lmin.txt
