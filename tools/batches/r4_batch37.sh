#!/bin/bash
# M = 256: where the run kernel overtakes the look-back tile kernel (FM and DeNo), finer than r4_batch34
cd /root/repo
for d in fm none; do
for nf in 8192 12288 16384 20480 24576; do
  for t in 1 1000000; do
    export CSDR_RUN_MIN_TILES=$t
    echo -n "$d nf=$nf min_tiles=$t: "; STEP_M=256 STEP_NF=$nf STEP_DEMOD=$d STEP_STEPS=300 timeout 300 python tools/step_time.py 2>&1 | grep -E "^region" | cut -c27-
  done
done
done
