#!/bin/bash
# round 5, batch 16: where the row-major k_agc_spec spends a 4096-frame call (TM_TRACE variant)
O=gpurun_out/r5q; mkdir -p $O
CSDR_LIB=$PWD/composable_sdr_amd/variants/libcsdr_tmtrace.so STEP_NF=4096 STEP_AGC=10 STEP_STEPS=2 python tools/step_time.py 2>&1 | grep -E "rm trace" | grep "wg 512" | tail -4 > $O/rmtrace.txt
cat $O/rmtrace.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "agc" 2>&1 | tail -3 > $O/tests.txt
cat $O/tests.txt
line() { python tools/step_time.py 2>&1 | grep -E '^no timer' | sed 's/no timer *: *//; s/;.*//'; }
for M in 64 256 1024; do
  echo "M=$M nf=4096 fm agc10: $(STEP_M=$M STEP_NF=4096 STEP_DEMOD=fm STEP_AGC=10 STEP_STEPS=400 line)" >> $O/sizes.txt
done
for nf in 1024 2048 8192 16384; do
  echo "M=256 nf=$nf fm agc10: $(STEP_M=256 STEP_NF=$nf STEP_DEMOD=fm STEP_AGC=10 STEP_STEPS=300 line)" >> $O/sizes.txt
done
cat $O/sizes.txt
