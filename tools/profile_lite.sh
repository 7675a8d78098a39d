#!/bin/bash
# Kernel trace + the two HBM traffic passes (FETCH_SIZE and WRITE_SIZE do not fit one pass; the pool refuses --pmc
# combined with tracing domains) of one bench.py configuration.  Usage (GPU box, repo root): tools/profile_lite.sh TAG [bench args]
set -u
TAG=${1:-run}; shift || true
export TMPDIR=/tmp
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
# keep only what tools/collect_profile.py reads (gpurun copies back at most 64 MiB)
prune() { find $OUT -type f ! -name "*kernel_stats.csv" ! -name "*counter_collection.csv" ! -name "*.log" ! -name "args.txt" -delete; find $OUT -name "*.log" -size +200k -delete; }
ARGS="--steps 5 --warmup 1 --no-cpu-baseline --no-other-configs --preheat-ms 100 $*"
rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace -o t -- python3 bench.py $ARGS > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "k_" -f csv -d $OUT/pmc3 -o p -- python3 bench.py $ARGS > $OUT/pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "k_" -f csv -d $OUT/pmc4 -o p -- python3 bench.py $ARGS > $OUT/pmc4.log 2>&1
echo "$ARGS" > $OUT/args.txt
grep -h '"metric"' $OUT/trace.log | head -1 | cut -c1-300
prune
