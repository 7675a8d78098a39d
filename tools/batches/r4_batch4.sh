#!/bin/bash
# round 4, batch 4: tile-major AGC tail (k_agc_spec_tm): bit identity, oracle parity, timing against the row-major route
O=gpurun_out/r4d; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q -s -k "tile_major or agc_tail_full_size or agc_whole_chunk or (run_sized and 10.0) or agc_tail_steady or chain_agc_fm_matches" > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
grep -E "passed|failed|rc=|tile-major|bit-identical|mismatch|Error|error|assert" $O/tests.log | tail -30
for i in 1 2; do for tm in 1 0; do
  echo "tm=$tm: $(CSDR_AGC_TM=$tm STEP_AGC=10 STEP_STEPS=300 timeout 300 python tools/step_time.py 2>&1 | grep -E '^(no timer|event pair)' | sed -e 's/ per step.*kernel/ kernel/' | tr '\n' ' ')" >> $O/time.txt
done; done
cat $O/time.txt
export TMPDIR=/tmp
for tm in 1 0; do
  CSDR_AGC_TM=$tm rocprofv3 --kernel-trace --stats -f csv -d $O/trace_tm$tm -o t -- python3 tools/kernel_time.py fm 256 262144 10 > $O/trace_tm$tm.log 2>&1
  python3 - $O/trace_tm$tm <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print(f"  {r['Name'][:70]:70s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.1f} us")
PY
done
find $O -name "*.csv" -size +2M -delete; find $O -type f -name "*.db" -delete
