#!/bin/bash
# Profiles bench.py on the GPU box: kernel trace + separate PMC passes (the pool refuses
# --pmc combined with tracing domains; FETCH_SIZE and WRITE_SIZE do not fit one pass).
# Usage (on the GPU box, from the repo root): tools/profile.sh TAG [bench args...]
set -u
TAG=${1:-run}; shift || true
export TMPDIR=/tmp
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
# keep only what tools/collect_profile.py reads (gpurun copies back at most 64 MiB)
prune() { find $OUT -type f ! -name "*kernel_stats.csv" ! -name "*counter_collection.csv" ! -name "*.log" ! -name "args.txt" -delete; find $OUT -name "*.log" -size +200k -delete; }
ARGS="--steps 5 --warmup 1 --no-cpu-baseline --no-agc-variant --preheat-ms 100 $*"
KRE='k_'
rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace -o t -- python3 bench.py $ARGS > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE \
   --kernel-include-regex "$KRE" -f csv -d $OUT/pmc1 -o p -- python3 bench.py $ARGS > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM \
   --kernel-include-regex "$KRE" -f csv -d $OUT/pmc2 -o p -- python3 bench.py $ARGS > $OUT/pmc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "$KRE" -f csv -d $OUT/pmc3 -o p -- python3 bench.py $ARGS > $OUT/pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "$KRE" -f csv -d $OUT/pmc4 -o p -- python3 bench.py $ARGS > $OUT/pmc4.log 2>&1
echo "$ARGS" > $OUT/args.txt
grep -h '"metric"' $OUT/trace.log | head -1
prune
