#!/bin/bash
# k_run1024v3 ablations (timing only): which role / phase paces the step
cd /root/repo
for v in default b3a2 b3a4 b3a16 b3a32 b3a64 b3a128 b3a80; do
  if [ $v = default ]; then L=""; else L="$PWD/composable_sdr_amd/variants/libcsdr_$v.so"; fi
  echo "== $v"
  CSDR_LIB=$L STEP_M=1024 STEP_STEPS=100 timeout 300 python tools/step_time.py 2>&1 | grep -E "^region"
done
