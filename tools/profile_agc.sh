#!/bin/bash
# SQ counter passes for the AGC tail kernel (bench.py's agc_variant leg).  Usage: tools/profile_agc.sh TAG
set -u
TAG=${1:-agc}; shift || true
export TMPDIR=/tmp
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
ARGS="--steps 3 --warmup 1 --no-cpu-baseline $*"
KRE='k_agc_spec'
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS \
   --kernel-include-regex "$KRE" -f csv -d $OUT/pmc1 -o p -- python3 bench.py $ARGS > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM SQ_INSTS_VALU_TRANS_F32 \
   --kernel-include-regex "$KRE" -f csv -d $OUT/pmc2 -o p -- python3 bench.py $ARGS > $OUT/pmc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "$KRE" -f csv -d $OUT/pmc3 -o p -- python3 bench.py $ARGS > $OUT/pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "$KRE" -f csv -d $OUT/pmc4 -o p -- python3 bench.py $ARGS > $OUT/pmc4.log 2>&1
