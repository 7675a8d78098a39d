#!/bin/bash
# round 5, batch 2: the collectives under the C ABI (world of one), the overlapped hybrid step, CSDR_FLAG_DFT_BACKWARD, the sharded C++ host;
# bench.py --shard channel --mix on one GPU through csdr_chain_process_device_mix; where SQ_LDS_BANK_CONFLICT of k_run256v2 comes from (no-DMA build)
O=gpurun_out/r5b; mkdir -p $O
python -m pytest tests -m gpu -x -q --durations=15 -k "comm_collectives or hybrid_step_overlapped or dft_direction or channel_shards_and_mix or bench_channel_shard or hybrid_sharding or cpp_soapy or run1024_v3_matches or mix_identity_equals" > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
tail -25 $O/tests.log
python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-agc-variant --preheat-ms 300 --shard channel --mix --channels 4096 --frames 16384 --demod none > $O/bench_mix_c.json 2> $O/bench_mix_c.err; tail -c 400 $O/bench_mix_c.err; cat $O/bench_mix_c.json
python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-agc-variant --preheat-ms 300 --shard channel --mix --channels 256 --demod fm > $O/bench_mix_256.json 2> $O/bench_mix_256.err; tail -c 400 $O/bench_mix_256.err; cat $O/bench_mix_256.json
python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-agc-variant --preheat-ms 300 > $O/bench_quick.json 2> $O/bench_quick.err; tail -c 300 $O/bench_quick.err; cat $O/bench_quick.json
V=$PWD/composable_sdr_amd/variants
CSDR_LIB=$V/libcsdr_abl1.so tools/pmc_quick.sh r5b_abl1 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" fm > $O/lds_abl1.txt 2>&1; cat $O/lds_abl1.txt
