#!/bin/bash
# k_run64v2 against k_run64 (CSDR_RUN64_V1=1) by call size, DeNo
cd /root/repo
for nf in 4096 16384 65536 131072 262144 1048576; do
  for v in 0 1; do
    if [ $v = 1 ]; then export CSDR_RUN64_V1=1; else unset CSDR_RUN64_V1; fi
    echo -n "nf=$nf v1=$v: "; STEP_M=64 STEP_NF=$nf STEP_DEMOD=none STEP_STEPS=200 timeout 300 python tools/step_time.py 2>&1 | grep -E "^region" | cut -c27-
  done
done
