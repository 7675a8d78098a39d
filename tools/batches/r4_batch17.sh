#!/bin/bash
O=gpurun_out/r4w; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "tile_major or agc_tail or (run_sized and 10.0)" > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log; tail -3 $O/tests.log
for L in 1568 2080 2592; do echo "== trace L=$L"; CSDR_LIB=$PWD/composable_sdr_amd/variants/libcsdr_tmtrace.so CSDR_AGC_L_TM=$L python tools/kernel_time.py fm 256 262144 10 2>&1 | grep -E "gain wave" | tail -2; done
for L in 0 1056 1568 2080 2592; do
  if [ $L = 0 ]; then unset CSDR_AGC_L_TM; else export CSDR_AGC_L_TM=$L; fi
  echo "L_TM=$L: $(STEP_AGC=10 STEP_STEPS=300 timeout 300 python tools/step_time.py 2>&1 | grep -E '^(no timer)' | sed -e 's/; kernel.*//' | tr '\n' ' ')" >> $O/sweep.txt
done
cat $O/sweep.txt
