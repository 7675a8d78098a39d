#!/bin/bash
# round 5, batch 4: k_run256v2 with the a & 7 run swizzle (y' write-back conflict-free): parity subset, SQ LDS counters, A/B against batch 1's build
O=gpurun_out/r5d; mkdir -p $O
python -m pytest tests -m gpu -x -q -k "fused256 or bench_layout_cfg3 or interleaved_shard or submit_device or tile_major or second_generation or smoke or chunk_invariance or agc_tail_full" > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
tail -4 $O/tests.log
V=$PWD/composable_sdr_amd/variants
for i in 1 2; do for v in default rsw0; do
  if [ $v = default ]; then L=""; else L="$V/libcsdr_$v.so"; fi
  echo "$v: $(CSDR_LIB=$L STEP_STEPS=800 python tools/step_time.py 2>&1 | grep -E '^(no timer|event pair)' | sed -e 's/ per step.*kernel/ kernel/' | tr '\n' ' ')" >> $O/rsw_ab.txt
done; done
cat $O/rsw_ab.txt
tools/pmc_quick.sh r5d_default "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" fm > $O/lds_default.txt 2>&1; cat $O/lds_default.txt
python bench.py --shard channel --mix --channels 4096 --frames 16384 --demod none --steps 5 --warmup 1 --no-cpu-baseline --preheat-ms 300 > $O/bench_mix_c.json 2> $O/bench_mix_c.err; wc -l $O/bench_mix_c.json
