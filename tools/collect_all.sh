#!/bin/bash
# After tools/profile_all.sh rNN on the GPU box: copy the summaries into profiles/ and refresh profiles/traffic.json
R=${1:-rXX}
python3 tools/collect_profile.py gpurun_out/prof_${R}_cfg3_fm ${R}_cfg3_fm 256 262144 primary > /dev/null
python3 tools/collect_profile.py gpurun_out/prof_${R}_cfg3_deno ${R}_cfg3_deno 256 262144 primary > /dev/null
python3 tools/collect_profile.py gpurun_out/prof_${R}_cfg3_agc ${R}_cfg3_agc 256 262144 tm > /dev/null
python3 tools/collect_profile.py gpurun_out/prof_${R}_cfg2_m64_deno ${R}_cfg2_m64_deno 64 1048576 > /dev/null
python3 tools/collect_profile.py gpurun_out/prof_${R}_cfg4shape_1024_fm_v3 ${R}_cfg4shape_1024_fm_v3 1024 65536 > /dev/null
python3 tools/collect_profile.py gpurun_out/prof_${R}_1024_deno_v3 ${R}_1024_deno_v3 1024 65536 > /dev/null
python3 tools/collect_profile.py gpurun_out/prof_${R}_cfg5shape_4096_mix ${R}_cfg5shape_4096_mix 4096 16384 > /dev/null
python3 tools/collect_profile.py gpurun_out/prof_${R}_4096_deno ${R}_4096_deno 4096 16384 > /dev/null
python3 tools/collect_profile.py gpurun_out/prof_${R}_4096_fm ${R}_4096_fm 4096 16384 > /dev/null
python3 tools/collect_profile.py gpurun_out/prof_${R}_4096_fm_mix ${R}_4096_fm_mix 4096 16384 > /dev/null
cat profiles/traffic.json | head -80
