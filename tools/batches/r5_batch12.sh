#!/bin/bash
# round 5, batch 12: AGC calls of the reference's chunk size: segment length sweep (row-major route) per M, kernel breakdown, short calls after the pilot fix
O=gpurun_out/r5l; mkdir -p $O
line() { python tools/step_time.py 2>&1 | grep -E '^no timer' | sed 's/no timer *: *//; s/;.*//'; }
for nf in 1024 2048 4096; do
  echo "M=256 nf=$nf fm agc10: $(STEP_M=256 STEP_NF=$nf STEP_DEMOD=fm STEP_AGC=10 STEP_STEPS=300 line)" >> $O/sizes.txt
done
for M in 256 1024 4096; do
  for L in 16 48 112 240 400 784; do
    echo "M=$M nf=4096 fm agc10 L=$L: $(CSDR_AGC_L=$L STEP_M=$M STEP_NF=4096 STEP_DEMOD=fm STEP_AGC=10 STEP_STEPS=200 line)" >> $O/sizes.txt
  done
done
cat $O/sizes.txt
export TMPDIR=/tmp
export STEP_DEMOD=fm STEP_AGC=10 STEP_STEPS=200 STEP_NF=4096
for M in 256 4096; do
  export STEP_M=$M
  rocprofv3 --kernel-trace --stats -d /tmp/prof_agc_$M -o agc -- python3 tools/step_time.py > $O/prof_agc_$M.log 2>&1
  python3 tools/rocpd_stats.py /tmp/prof_agc_$M/agc_results.db > $O/agc4096frames_M${M}_kernel_stats.txt 2>&1
  head -8 $O/agc4096frames_M${M}_kernel_stats.txt | cut -c1-220
done
du -sh gpurun_out
