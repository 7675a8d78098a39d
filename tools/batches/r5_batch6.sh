#!/bin/bash
# round 5, batch 6: fused 4096 route: parity + per-kernel times (rocprofv3 kernel trace; raw traces deleted: gpurun_out must stay small)
O=gpurun_out/r5f; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "fused4096" > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
tail -12 $O/tests.log
export TMPDIR=/tmp
for d in none fm; do
rocprofv3 --kernel-trace --stats -f csv -d /tmp/prof_$d -o p -- python3 tools/kernel_time.py $d 4096 16384 > $O/prof_$d.log 2>&1
python3 - /tmp/prof_$d > $O/kstats_$d.txt <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "csdr" in r["Name"] or "rocclr" in r["Name"]: print(r["Name"][:90], r["Calls"], r["AverageNs"])
PY
cat $O/kstats_$d.txt; tail -1 $O/prof_$d.log
done
python tools/kernel_time.py fm 4096 16384 0 mix 2>&1 | tail -1
python tools/kernel_time.py fm 4096 16384 10 2>&1 | tail -1
python tools/kernel_time.py none 4096 4096 2>&1 | tail -1
