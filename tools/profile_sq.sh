#!/bin/bash
# Extra SQ counter passes for the run kernels (issue mix, LDS stalls, instruction fetch).  Usage: tools/profile_sq.sh TAG [bench args]
set -u
TAG=${1:-sq}; shift || true
export TMPDIR=/tmp
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
ARGS="--steps 5 --warmup 1 --no-cpu-baseline --no-agc-variant $*"
KRE='k_run256|k_run64'
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_THREAD_CYCLES_VALU \
   --kernel-include-regex "$KRE" -f csv -d $OUT/pmc5 -o p -- python3 bench.py $ARGS > $OUT/pmc5.log 2>&1
rocprofv3 --pmc SQ_LDS_ADDR_CONFLICT SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_UNALIGNED_STALL SQ_INST_LEVEL_LDS SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_LDS_IDX_ACTIVE \
   --kernel-include-regex "$KRE" -f csv -d $OUT/pmc6 -o p -- python3 bench.py $ARGS > $OUT/pmc6.log 2>&1
rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_LEVEL_VMEM \
   --kernel-include-regex "$KRE" -f csv -d $OUT/pmc7 -o p -- python3 bench.py $ARGS > $OUT/pmc7.log 2>&1
