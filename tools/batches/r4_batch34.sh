#!/bin/bash
# M = 256 FM: run kernel (CSDR_RUN_MIN_TILES=1) against the look-back tile kernel (CSDR_RUN_MIN_TILES=1000000) by call size
cd /root/repo
for nf in 4096 16384 32768 65536 131072; do
  for t in 1 1000000 default; do
    if [ $t = default ]; then unset CSDR_RUN_MIN_TILES; else export CSDR_RUN_MIN_TILES=$t; fi
    echo -n "nf=$nf min_tiles=$t: "; STEP_M=256 STEP_NF=$nf STEP_STEPS=200 timeout 300 python tools/step_time.py 2>&1 | grep -E "^region" | cut -c27-
  done
done
