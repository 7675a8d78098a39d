#!/bin/bash
# round 5, batch 14: k_agc_spec_tm with a mover wave (tile DMA out of the gain wave): bit identity, then time against the two-wave build
O=gpurun_out/r5o; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "agc_tail or cfg3_256ch_fm_agc" 2>&1 | tail -5 > $O/tests.txt
cat $O/tests.txt
for v in "" mov0; do
  L=""; [ -n "$v" ] && L=$PWD/composable_sdr_amd/variants/libcsdr_$v.so
  for i in 1 2; do
    echo "variant '${v:-product}': $(CSDR_LIB=$L STEP_AGC=10 STEP_STEPS=200 python tools/step_time.py 2>&1 | grep -E '^no timer' | sed 's/;.*//')" >> $O/times.txt
  done
done
cat $O/times.txt
CSDR_LIB=$PWD/composable_sdr_amd/variants/libcsdr_tmtrace.so STEP_AGC=10 STEP_STEPS=2 python tools/step_time.py 2>&1 | grep -E "tm trace" | grep "wg 336" | head -4 > $O/tmtrace.txt
cat $O/tmtrace.txt
