#!/bin/bash
# reference-sized chunks (4096 frames per call = 4 M 1024 samples): kernel and step time per call
cd /root/repo
for M in 64 256 1024; do
  for d in fm none; do
    echo "== M=$M demod=$d nf=4096"
    STEP_M=$M STEP_NF=4096 STEP_DEMOD=$d STEP_STEPS=500 timeout 300 python tools/step_time.py 2>&1 | grep -E "^region|^no timer"
  done
done
echo "== M=256 fm agc nf=4096"; STEP_M=256 STEP_NF=4096 STEP_AGC=10 STEP_STEPS=300 timeout 300 python tools/step_time.py 2>&1 | grep -E "^no timer"
