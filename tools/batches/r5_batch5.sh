#!/bin/bash
# round 5, batch 5: first runs of the fused 4096-channel route
O=gpurun_out/r5e; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "fused4096 or config5 or mix_identity_equals or 4096" > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
tail -30 $O/tests.log
for d in none fm; do
timeout 300 python tools/kernel_time.py $d 4096 16384 > $O/kt_$d.txt 2>&1; tail -3 $O/kt_$d.txt
done
