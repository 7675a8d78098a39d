#!/bin/bash
# round 4, batch 8: where k_agc_spec_tm's block time goes (timing-only ablations)
O=gpurun_out/r4h; mkdir -p $O
for L in 1568 2592; do for v in default tmabl1 tmabl2 tmabl3 tmabl4 tmabl7; do
  if [ $v = default ]; then LIB=""; else LIB="$PWD/composable_sdr_amd/variants/libcsdr_$v.so"; fi
  echo "L=$L $v: $(CSDR_LIB=$LIB CSDR_AGC_L_TM=$L STEP_AGC=10 STEP_STEPS=200 timeout 300 python tools/step_time.py 2>&1 | grep -E '^(no timer)' | sed -e 's/; kernel.*//' | tr '\n' ' ')" >> $O/abl.txt
done; done
cat $O/abl.txt
