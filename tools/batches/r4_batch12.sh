#!/bin/bash
O=gpurun_out/r4l; mkdir -p $O
for v in default spread1 spread2; do for L in 1568 2080; do
  if [ $v = default ]; then LIB=""; else LIB="$PWD/composable_sdr_amd/variants/libcsdr_$v.so"; fi
  echo "$v L=$L: $(CSDR_LIB=$LIB CSDR_AGC_L_TM=$L STEP_AGC=10 STEP_STEPS=200 timeout 300 python tools/step_time.py 2>&1 | grep -E '^(no timer)' | sed -e 's/; kernel.*//' | tr '\n' ' ')" >> $O/spread.txt
done; done
cat $O/spread.txt
