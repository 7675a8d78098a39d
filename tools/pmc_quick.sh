#!/bin/bash
# One-off counter passes over tools/kernel_time.py (GPU box, repo root): tools/pmc_quick.sh TAG "CTR1 CTR2" [kernel_time args]
# Each counter gets its own pass (no tracing domains next to --pmc on this pool); prints the per-dispatch median per kernel.
set -u
TAG=$1; CTRS=$2; shift 2
export TMPDIR=/tmp
OUT=gpurun_out/pmcq_$TAG
mkdir -p $OUT
for c in $CTRS; do
  rocprofv3 --pmc $c --kernel-include-regex "k_run|k_agc|k_dc|k_pfb" -f csv -d $OUT/$c -o p -- python3 tools/kernel_time.py "$@" > $OUT/$c.log 2>&1
  python3 - "$OUT/$c" "$c" <<'PY'
import csv, glob, sys, collections
d, c = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == c:
            acc[r["Kernel_Name"].split("(")[0][:60]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    v.sort()
    print(f"{c:28s} {k:60s} median {v[len(v)//2]:16.0f}  n={len(v)}")
PY
done
find $OUT -type f ! -name "*.log" -delete
