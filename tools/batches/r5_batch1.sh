#!/bin/bash
# round 5, batch 1: padded LDS image of k_run256v2 (V2_PAD=1, verdict r04 #1a): full GPU suite with durations, A/B against the dense image,
# SQ LDS conflict counters of both, the no-DMA / no-store ablations on today's kernel (verdict r04 #1c), run weights
O=gpurun_out/r5a; mkdir -p $O
python -m pytest tests -m gpu -x -q --durations=70 > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
tail -4 $O/tests.log
V=$PWD/composable_sdr_amd/variants
for i in 1 2; do for v in default pad0; do
  if [ $v = default ]; then L=""; else L="$V/libcsdr_$v.so"; fi
  echo "$v: $(CSDR_LIB=$L STEP_STEPS=800 python tools/step_time.py 2>&1 | grep -E '^(no timer|event pair)' | sed -e 's/ per step.*kernel/ kernel/' | tr '\n' ' ')" >> $O/pad_ab.txt
done; done
cat $O/pad_ab.txt
for v in default pad0; do
  if [ $v = default ]; then L=""; else L="$V/libcsdr_$v.so"; fi
  CSDR_LIB=$L tools/pmc_quick.sh r5a_$v "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" fm > $O/lds_$v.txt 2>&1
done
cat $O/lds_*.txt
for v in default abl1 abl2 abl3 abl62; do
  if [ $v = default ]; then L=""; else L="$V/libcsdr_$v.so"; fi
  echo "== $v" >> $O/abl.txt
  CSDR_LIB=$L STEP_STEPS=400 python tools/step_time.py 2>&1 | grep -E "^region|^no timer" >> $O/abl.txt
done
cat $O/abl.txt
for w in "1.2,0.8" "1.24,0.76" "1.28,0.72"; do
  echo "== weights $w" >> $O/w.txt
  CSDR_RUN_WEIGHTS=$w STEP_STEPS=400 python tools/step_time.py 2>&1 | grep -E "^region" >> $O/w.txt
done
cat $O/w.txt
