#!/bin/bash
# round 5, batch 23: LLVM scheduling strategies (max-memory-clause / max-ilp) on the other hot translation units, event-pair kernel times
O=gpurun_out/r6c; mkdir -p $O
t() { python tools/step_time.py 2>&1 | grep -E '^event pair' | sed 's/event pair per launch *: *//'; }
run() {  # tag, env...
  for v in "" ${1}_memory ${1}_ilp ""; do
    L=""; [ -n "$v" ] && L=$PWD/composable_sdr_amd/variants/libcsdr_$v.so
    echo "$2 '${v:-product}': $(env CSDR_LIB=$L $3 STEP_STEPS=200 python tools/step_time.py 2>&1 | grep -E "^$4" | sed 's/^[a-z ]*: *//')" >> $O/sched.txt
  done
}
run run1024_v3 "M=1024 fm" "STEP_M=1024 STEP_DEMOD=fm" "event pair"
run run1024_v3 "M=1024 none" "STEP_M=1024 STEP_DEMOD=none" "event pair"
run pfb4096 "M=4096 none" "STEP_M=4096 STEP_DEMOD=none" "no timer"
run pfb4096 "M=4096 fm" "STEP_M=4096 STEP_DEMOD=fm" "no timer"
run agc_tail "M=256 fm agc" "STEP_M=256 STEP_AGC=10" "no timer"
run run64_v2 "M=64 none" "STEP_M=64 STEP_DEMOD=none" "event pair"
cat $O/sched.txt
