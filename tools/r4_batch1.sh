#!/bin/bash
# round 4, first GPU batch: new parity tests, the restructured bench line, the selective cache-policy A/B (verdict r03 #1 (ii)), phase trace
O=gpurun_out/r4a; mkdir -p $O
python -m pytest tests -m gpu -x -q -k "run_sized or oracle_rows or short_chunks or agc_whole_chunk or bench_channel_shard or submit_device or round3_entry" > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
tail -5 $O/tests.log
python bench.py > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.err
for i in 1 2; do for v in default sel_nt sel_sc1 sel_sc01 sel_sc1nt sel_sc01nt; do
  if [ $v = default ]; then L=""; else L="$PWD/composable_sdr_amd/variants/libcsdr_$v.so"; fi
  echo "$v: $(CSDR_LIB=$L STEP_STEPS=800 python tools/step_time.py 2>&1 | grep -E '^(no timer|event pair)' | sed -e 's/ per step.*kernel/ kernel/' | tr '\n' ' ')" >> $O/policy.txt
done; done
cat $O/policy.txt
for v in default sel_nt sel_sc01nt; do
  if [ $v = default ]; then L=""; else L="$PWD/composable_sdr_amd/variants/libcsdr_$v.so"; fi
  CSDR_LIB=$L tools/pmc_quick.sh r4a_$v "FETCH_SIZE" fm > $O/fetch_$v.txt 2>&1
done
cat $O/fetch_*.txt
CSDR_TRACE=1 python tools/trace_tiles.py > $O/trace1.txt 2>&1; cat $O/trace1.txt
CSDR_TRACE=2 python tools/trace_tiles.py > $O/trace2.txt 2>&1; cat $O/trace2.txt
