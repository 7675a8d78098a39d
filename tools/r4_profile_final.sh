#!/bin/bash
# Final profile pass of round 4 (every configuration a number is quoted for, with the library as committed); the cfg4-shape tag of
# tools/profile_all.sh keeps round 4's k_run1024v2<FM> evidence, k_run1024v3 is profiled under its own tags (FM with the SQ counter passes)
R=r04
tools/profile.sh ${R}_cfg3_fm --no-agc-variant
tools/profile_lite.sh ${R}_cfg3_deno --demod none --no-agc-variant
tools/profile_lite.sh ${R}_cfg3_agc --steps 3
tools/profile_lite.sh ${R}_cfg2_m64_deno --channels 64 --frames 1048576 --demod none --no-agc-variant
tools/profile_lite.sh ${R}_cfg5shape_4096_mix --channels 4096 --frames 16384 --demod none --mix --no-agc-variant
tools/profile.sh ${R}_cfg4shape_1024_fm_v3 --channels 1024 --frames 65536
tools/profile_lite.sh ${R}_1024_deno_v3 --channels 1024 --frames 65536 --demod none --no-agc-variant
