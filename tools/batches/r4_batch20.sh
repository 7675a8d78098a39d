#!/bin/bash
O=gpurun_out/r4z; mkdir -p $O
for i in 1 2; do for e in 0 1; do
  if [ $e = 0 ]; then unset CSDR_EXP_FM_PAIR_MAJOR; else export CSDR_EXP_FM_PAIR_MAJOR=1; fi
  echo "pair-major=$e: $(STEP_STEPS=600 timeout 300 python tools/step_time.py 2>&1 | grep -E '^(no timer|event pair)' | sed -e 's/ per step.*kernel/ kernel/' | tr '\n' ' ')" >> $O/pm.txt
done; done
cat $O/pm.txt
unset CSDR_EXP_FM_PAIR_MAJOR
for e in 0 1; do if [ $e = 0 ]; then unset CSDR_EXP_FM_PAIR_MAJOR; else export CSDR_EXP_FM_PAIR_MAJOR=1; fi; echo "== pair-major=$e" >> $O/power.txt; POWER=1 POWER_SECONDS=4 STEP_STEPS=50 timeout 300 python tools/step_time.py 2>&1 | grep -E "smi|sustained" | sed -e "s/'Temperature[^,]*, //" -e "s/'fclk[^,]*, //g" -e "s/'mclk[^,]*, //g" -e "s/'sclk clock level:[^,]*, //" | tail -4 >> $O/power.txt; done; cat $O/power.txt
