#!/bin/bash
# k_run1024v3 against k_run1024 (CSDR_RUN1024_V1=1) by call size, FM and DeNo
cd /root/repo
for d in fm none; do
for nf in 256 1024 4096 16384; do
  for v in 0 1; do
    if [ $v = 1 ]; then export CSDR_RUN1024_V1=1; else unset CSDR_RUN1024_V1; fi
    echo -n "$d nf=$nf v1=$v: "; STEP_M=1024 STEP_NF=$nf STEP_DEMOD=$d STEP_STEPS=200 timeout 300 python tools/step_time.py 2>&1 | grep -E "^region" | cut -c27-
  done
done
done
