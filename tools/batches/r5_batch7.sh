#!/bin/bash
# round 5, batch 7: fused 4096 route: parity (all cases) + the freqdem denormal fix + front-kernel ablations and run-count sweep
O=gpurun_out/r5g; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "fused4096 or collapsed_agc or freqdem or agc_tail_is_bit or chain_agc_fm" > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
tail -6 $O/tests.log
V=$PWD/composable_sdr_amd/variants
for v in default h1 h2 h4 h8 h15; do
  if [ $v = default ]; then L=""; else L="$V/libcsdr_$v.so"; fi
  echo "$v: $(CSDR_LIB=$L python tools/kernel_time.py none 4096 16384 2>&1 | tail -1 | sed 's/.*CF32>//')" >> $O/abl.txt
done
for r in 32 64 128; do echo "runs $r: $(CSDR_RUN4096_RUNS=$r python tools/kernel_time.py none 4096 16384 2>&1 | tail -1 | sed 's/.*CF32>//')" >> $O/abl.txt; done
cat $O/abl.txt
