#!/bin/bash
# round 5, batch 22: LLVM scheduling strategies on kernels_fused_v2.hip (timing only: three of them spill SGPRs, which the asm stores do not survive)
O=gpurun_out/r6b; mkdir -p $O
for v in "" sch_ilp sch_memc sch_iter sch_minreg ""; do
  L=""; [ -n "$v" ] && L=$PWD/composable_sdr_amd/variants/libcsdr_$v.so
  echo "variant '${v:-product}': $(CSDR_LIB=$L STEP_STEPS=300 python tools/step_time.py 2>&1 | grep -E '^event pair' | sed 's/event pair per launch *: *//')" >> $O/sched.txt
done
cat $O/sched.txt
