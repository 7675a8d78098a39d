#!/bin/bash
# round 5, batch 13: where k_agc_spec_tm's two waves spend their cycles (TM_TRACE variant, device printf at the end of three workgroups)
O=gpurun_out/r5n; mkdir -p $O
CSDR_LIB=$PWD/composable_sdr_amd/variants/libcsdr_tmtrace.so STEP_AGC=10 STEP_STEPS=2 python tools/step_time.py 2>&1 | grep -E "tm trace|per step" | grep "wg 336" | head -6 > $O/tmtrace.txt
cat $O/tmtrace.txt
