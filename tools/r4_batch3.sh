#!/bin/bash
# round 4, batch 3: k_run256v3 parity; V2_COLSCAN variant (LDS pass removal, verdict r03 #1 (iii)) parity + timing + power
O=gpurun_out/r4c; mkdir -p $O
V=$PWD/composable_sdr_amd/variants/libcsdr_colscan.so
timeout 900 python -m pytest tests -m gpu -x -q -k "run256_v3 or bench_channel_shard" > $O/tests_v3.log 2>&1; echo "tests rc=$?" >> $O/tests_v3.log
grep -E "passed|failed|rc=|vs v2|Error|error|assert" $O/tests_v3.log | tail -20
CSDR_LIB=$V timeout 900 python -m pytest tests -m gpu -x -q -s -k "fused256 or bench_layout_cfg3_256ch_fm_whole or full_size_cfg3 or submit_device_independent" > $O/tests_colscan.log 2>&1; echo "tests rc=$?" >> $O/tests_colscan.log
grep -E "passed|failed|rc=|rel-rms|median|Error|error|assert" $O/tests_colscan.log | tail -20
for i in 1 2 3; do for v in default colscan; do
  if [ $v = default ]; then L=""; else L="$V"; fi
  echo "$v: $(CSDR_LIB=$L STEP_STEPS=800 timeout 300 python tools/step_time.py 2>&1 | grep -E '^(no timer|event pair)' | sed -e 's/ per step.*kernel/ kernel/' | tr '\n' ' ')" >> $O/time.txt
done; done
cat $O/time.txt
for v in default colscan; do if [ $v = default ]; then L=""; else L="$V"; fi; echo "== $v" >> $O/power.txt; CSDR_LIB=$L POWER=1 POWER_SECONDS=4 STEP_STEPS=50 timeout 300 python tools/step_time.py 2>&1 | grep -E "smi|sustained" | sed -e "s/'Temperature[^,]*, //" -e "s/'fclk[^,]*, //g" -e "s/'mclk[^,]*, //g" -e "s/'sclk clock level:[^,]*, //" | tail -6 >> $O/power.txt; done
cat $O/power.txt
CSDR_LIB=$V CSDR_TRACE=1 timeout 300 python tools/trace_tiles.py > $O/trace_colscan.txt 2>&1; head -20 $O/trace_colscan.txt
