#!/bin/bash
# round 4, batch 6: k_agc_spec_tm segment-length sweep (odd multiples of 32) + kernel-level times
O=gpurun_out/r4f; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "(run_sized and 10.0)" > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log; tail -3 $O/tests.log
for L in 1056 1312 1440 1568 1696 1824 1952 2080 2336 2592; do
  echo "L_TM=$L: $(CSDR_AGC_L_TM=$L STEP_AGC=10 STEP_STEPS=300 timeout 300 python tools/step_time.py 2>&1 | grep -E '^(no timer)' | sed -e 's/; kernel.*//' | tr '\n' ' ')" >> $O/sweep.txt
done
cat $O/sweep.txt
export TMPDIR=/tmp
for L in 1568 2080; do
  CSDR_AGC_L_TM=$L rocprofv3 --kernel-trace --stats -f csv -d $O/trace_L$L -o t -- python3 tools/kernel_time.py fm 256 262144 10 > $O/trace_L$L.log 2>&1
  echo "== L=$L"; python3 - $O/trace_L$L <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "csdr" in r["Name"]: print(f"  {r['Name'][:70]:70s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.1f} us")
PY
done
find $O -name "*.csv" -size +2M -delete; find $O -type f -name "*.db" -delete
