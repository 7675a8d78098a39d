#!/bin/bash
# round 5, batch 24: max-memory-clause on kernels_fused_v2.hip, alternating A/B, 600 launches each
O=gpurun_out/r6d; mkdir -p $O
for i in 1 2 3; do
  for v in "" sch_memc; do
    L=""; [ -n "$v" ] && L=$PWD/composable_sdr_amd/variants/libcsdr_$v.so
    echo "pass $i '${v:-product}': $(CSDR_LIB=$L STEP_STEPS=600 python tools/step_time.py 2>&1 | grep -E '^event pair|^no timer' | sed 's/ per step.*kernel/ kernel/; s/ per step.*//' | tr '\n' ' ')" >> $O/ab.txt
  done
done
cat $O/ab.txt
