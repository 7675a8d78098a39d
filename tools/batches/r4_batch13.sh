#!/bin/bash
O=gpurun_out/r4m; mkdir -p $O
export TMPDIR=/tmp
run() { # tag counters...
  tag=$1; shift
  CSDR_AGC_L_TM=1568 rocprofv3 --pmc "$@" --kernel-include-regex "k_agc_spec" -f csv -d $O/$tag -o p -- python3 tools/kernel_time.py fm 256 262144 10 > $O/$tag.log 2>&1
  python3 - $O/$tag <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    v.sort(); print(f"  {k:28s} median {v[len(v)//2]:16.0f}  n={len(v)}")
PY
}
run a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY
run b SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY
run c SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_WAVE32_LDS SQ_THREAD_CYCLES_VALU
find $O -name "*.csv" -size +1M -delete; find $O -type f -name "*.db" -delete
