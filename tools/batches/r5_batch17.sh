#!/bin/bash
# round 5, batch 17: row-major k_agc_spec packing the segments of several channels into a workgroup: bit identity, then many-channel calls
O=gpurun_out/r5r; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "agc or fused4096" 2>&1 | tail -5 > $O/tests.txt
cat $O/tests.txt
line() { python tools/step_time.py 2>&1 | grep -E '^no timer' | sed 's/no timer *: *//; s/;.*//'; }
for M in 64 256 1024 4096; do
  echo "M=$M nf=4096 fm agc10: $(STEP_M=$M STEP_NF=4096 STEP_DEMOD=fm STEP_AGC=10 STEP_STEPS=300 line)" >> $O/sizes.txt
done
echo "M=4096 nf=16384 fm agc10: $(STEP_M=4096 STEP_NF=16384 STEP_DEMOD=fm STEP_AGC=10 STEP_STEPS=100 line)" >> $O/sizes.txt
echo "M=4096 nf=16384 none agc10: $(STEP_M=4096 STEP_NF=16384 STEP_DEMOD=none STEP_AGC=10 STEP_STEPS=100 line)" >> $O/sizes.txt
echo "M=1024 nf=1024 fm agc10: $(STEP_M=1024 STEP_NF=1024 STEP_DEMOD=fm STEP_AGC=10 STEP_STEPS=300 line)" >> $O/sizes.txt
echo "M=256 nf=1024 fm agc10: $(STEP_M=256 STEP_NF=1024 STEP_DEMOD=fm STEP_AGC=10 STEP_STEPS=300 line)" >> $O/sizes.txt
cat $O/sizes.txt
