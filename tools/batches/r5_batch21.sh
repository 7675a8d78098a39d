#!/bin/bash
# round 5, batch 21: whole GPU suite, smoke, every profile again (sources changed), default bench, M = 4096 + AGC at the reference chunk
O=gpurun_out/r6k; mkdir -p $O
( time timeout 1500 python -m pytest tests -q -m gpu ) > $O/tests.txt 2>&1
tail -6 $O/tests.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
line() { python tools/step_time.py 2>&1 | grep -E '^no timer' | sed 's/no timer *: *//; s/;.*//'; }
echo "M=4096 nf=4096 fm agc10: $(STEP_M=4096 STEP_NF=4096 STEP_DEMOD=fm STEP_AGC=10 STEP_STEPS=300 line)" >> $O/sizes.txt
echo "M=1024 nf=4096 fm agc10: $(STEP_M=1024 STEP_NF=4096 STEP_DEMOD=fm STEP_AGC=10 STEP_STEPS=300 line)" >> $O/sizes.txt
echo "M=256 nf=4096 fm agc10: $(STEP_M=256 STEP_NF=4096 STEP_DEMOD=fm STEP_AGC=10 STEP_STEPS=300 line)" >> $O/sizes.txt
cat $O/sizes.txt
tools/profile_all.sh r05 > $O/profile_all.log 2>&1
tail -3 $O/profile_all.log
python bench.py > $O/bench.json 2> $O/bench.err; cat $O/bench.json | cut -c1-1500
du -sh gpurun_out
