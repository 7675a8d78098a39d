#!/bin/bash
# round 5, batch 9: k_run256v2 without warm-up windows (RunArgs::nowu + k_run256_dcfix): parity, A/B against CSDR_NOWU=0
O=gpurun_out/r5i; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q -k "fused256 or bench_layout_cfg3 or submit_device or tile_major or second_generation or full_size_cfg3 or chunk_invariance or agc_tail_full or golden or smoke or chain_deno or chain_fm" > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
tail -5 $O/tests.log
for i in 1 2; do for v in 1 0; do
  echo "nowu=$v: $(CSDR_NOWU=$v STEP_STEPS=800 python tools/step_time.py 2>&1 | grep -E '^(no timer|event pair)' | sed -e 's/ per step.*kernel/ kernel/' | tr '\n' ' ')" >> $O/nowu_ab.txt
done; done
for v in 1 0; do echo "DeNo nowu=$v: $(CSDR_NOWU=$v STEP_DEMOD=none STEP_STEPS=400 python tools/step_time.py 2>&1 | grep -E '^(no timer)' | sed -e 's/;.*//')" >> $O/nowu_ab.txt; done
cat $O/nowu_ab.txt
