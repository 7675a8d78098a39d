#!/bin/bash
# The per-rank kernels of a channel-shard run (north_star's partition; BASELINE configs[3] per rank = 1024 channels, stride 8), each alone on
# this GPU through `bench.py --chan-stride G`: kernel trace + HBM traffic passes.  Usage (GPU box, repo root): tools/profile_shards.sh rNN ;
# then tools/collect_shards.sh rNN here.
R=${1:-rXX}
tools/profile.sh ${R}_shard_g8_m1024_fm --channels 1024 --frames 65536 --chan-stride 8
tools/profile_lite.sh ${R}_shard_g4_m1024_fm --channels 1024 --frames 65536 --chan-stride 4
tools/profile_lite.sh ${R}_shard_g8_m1024_fm_agc --channels 1024 --frames 65536 --chan-stride 8 --agc 10 --steps 3
tools/profile_lite.sh ${R}_shard_g8_m256_fm --chan-stride 8
tools/profile_lite.sh ${R}_shard_g8_m256_fm_agc --chan-stride 8 --agc 10 --steps 3
tools/profile_lite.sh ${R}_shard_g8_m4096_deno_mix --channels 4096 --frames 16384 --demod none --mix --chan-stride 8 --no-agc-variant
