#!/bin/bash
# round 5, batch 19: whole GPU suite, smoke, every profile again (the AGC tail sources changed: traffic.json is keyed by the source hash), default bench
O=gpurun_out/r5t; mkdir -p $O
( time timeout 1500 python -m pytest tests -x -q -m gpu ) > $O/tests.txt 2>&1
tail -6 $O/tests.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
tools/profile_all.sh r05 > $O/profile_all.log 2>&1
tail -3 $O/profile_all.log
python bench.py > $O/bench.json 2> $O/bench.err; cat $O/bench.json
du -sh gpurun_out
