#!/bin/bash
cd /root/repo
for v in default b3a8; do
  if [ $v = default ]; then L=""; else L="$PWD/composable_sdr_amd/variants/libcsdr_$v.so"; fi
  echo "== $v"
  CSDR_LIB=$L STEP_M=1024 STEP_STEPS=200 timeout 300 python tools/step_time.py 2>&1 | grep -E "^region"
done
CSDR_LIB=$PWD/composable_sdr_amd/variants/libcsdr_b3a8.so timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "run1024_v3" 2>&1 | tail -2
