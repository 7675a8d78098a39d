#!/bin/bash
# k_agc_spec_tm: wave roles dealt by placement (SIMD number + wave slot) against wave 0 = gain; bit identity first
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "tile_major_route or agc_tail_steady or bench_layout_cfg3_256ch_fm_agc" 2>&1 | tail -3
for v in default tmnoroles default tmnoroles; do
  if [ $v = default ]; then L=""; else L="$PWD/composable_sdr_amd/variants/libcsdr_$v.so"; fi
  echo -n "$v: "; CSDR_LIB=$L STEP_AGC=10 STEP_STEPS=300 timeout 300 python tools/step_time.py 2>&1 | grep -E "^no timer" | cut -c27-
done
