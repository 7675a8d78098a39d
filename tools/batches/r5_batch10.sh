#!/bin/bash
# round 5, batch 10: run weights / run-kernel threshold with the cheaper run start; then every profile DESIGN / README quote (tools/profile_all.sh r05)
O=gpurun_out/r5j; mkdir -p $O
for w in "1.2,0.8" "1.24,0.76" "1.28,0.72" "1.16,0.84"; do
  echo "weights $w: $(CSDR_RUN_WEIGHTS=$w STEP_STEPS=400 python tools/step_time.py 2>&1 | grep -E '^region' | sed 's/;.*//')" >> $O/w.txt
done
cat $O/w.txt
for nf in 8192 12288 16384; do
  echo "nf=$nf run / tile: $(CSDR_RUN_MIN_TILES=1 STEP_NF=$nf STEP_STEPS=400 python tools/step_time.py 2>&1 | grep -E '^no timer' | sed 's/ us per.*//; s/no timer *: *//') / $(CSDR_RUN_MIN_TILES=100000 STEP_NF=$nf STEP_STEPS=400 python tools/step_time.py 2>&1 | grep -E '^no timer' | sed 's/ us per.*//; s/no timer *: *//')" >> $O/thr.txt
done
cat $O/thr.txt
tools/profile_all.sh r05 > $O/profile_all.log 2>&1
tail -3 $O/profile_all.log
du -sh gpurun_out
