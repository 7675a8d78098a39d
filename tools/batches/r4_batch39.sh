#!/bin/bash
# the 1024-tile threshold with the AGC on: mid-sized calls (run kernel + tile-major tail) against the previous choice (tile kernel + row-major tail)
cd /root/repo
for nf in 16384 24576 32768; do for t in default 2048; do
  if [ $t = default ]; then unset CSDR_RUN_MIN_TILES; else export CSDR_RUN_MIN_TILES=$t; fi
  echo -n "agc nf=$nf min_tiles=$t: "; STEP_M=256 STEP_NF=$nf STEP_AGC=10 STEP_STEPS=200 timeout 300 python tools/step_time.py 2>&1 | grep -E "^no timer" | cut -c27-
done; done
