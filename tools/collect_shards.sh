#!/bin/bash
# After tools/profile_shards.sh rNN on the GPU box: copy the summaries into profiles/ and refresh profiles/traffic.json
R=${1:-rXX}
python3 tools/collect_profile.py gpurun_out/prof_${R}_shard_g8_m1024_fm ${R}_shard_g8_m1024_fm 1024 65536 primary > /dev/null
python3 tools/collect_profile.py gpurun_out/prof_${R}_shard_g4_m1024_fm ${R}_shard_g4_m1024_fm 1024 65536 primary > /dev/null
python3 tools/collect_profile.py gpurun_out/prof_${R}_shard_g8_m1024_fm_agc ${R}_shard_g8_m1024_fm_agc 1024 65536 > /dev/null
python3 tools/collect_profile.py gpurun_out/prof_${R}_shard_g8_m256_fm ${R}_shard_g8_m256_fm 256 262144 primary > /dev/null
python3 tools/collect_profile.py gpurun_out/prof_${R}_shard_g8_m256_fm_agc ${R}_shard_g8_m256_fm_agc 256 262144 > /dev/null
python3 tools/collect_profile.py gpurun_out/prof_${R}_shard_g8_m4096_deno_mix ${R}_shard_g8_m4096_deno_mix 4096 16384 primary > /dev/null
grep -n 'G8\|G4\|fold8' profiles/traffic.json | head
