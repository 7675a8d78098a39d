#!/bin/bash
# Every configuration DESIGN.md / README.md quote a number for: kernel trace + HBM traffic passes, the headline and the 1024-channel FM
# kernel also with the SQ counter passes.  Usage (GPU box, repo root): tools/profile_all.sh rNN ; then tools/collect_all.sh rNN here.
# (profiles/r04_cfg4shape_1024_fm_* is round 4's k_run1024v2<FM> evidence: CSDR_RUN1024_V3=0 tools/profile_lite.sh ... reproduces it.)
R=${1:-rXX}
tools/profile.sh ${R}_cfg3_fm
tools/profile_lite.sh ${R}_cfg3_deno --demod none --no-agc-variant
tools/profile_lite.sh ${R}_cfg3_agc --steps 3
tools/profile_lite.sh ${R}_cfg2_m64_deno --channels 64 --frames 1048576 --demod none --no-agc-variant
tools/profile.sh ${R}_cfg4shape_1024_fm_v3 --channels 1024 --frames 65536
tools/profile_lite.sh ${R}_1024_deno_v3 --channels 1024 --frames 65536 --demod none --no-agc-variant
tools/profile_lite.sh ${R}_cfg5shape_4096_mix --channels 4096 --frames 16384 --demod none --mix --no-agc-variant
tools/profile_lite.sh ${R}_4096_deno --channels 4096 --frames 16384 --demod none --no-agc-variant
tools/profile_lite.sh ${R}_4096_fm --channels 4096 --frames 16384 --demod fm --no-agc-variant
tools/profile_lite.sh ${R}_4096_fm_mix --channels 4096 --frames 16384 --demod fm --mix --no-agc-variant
