#!/bin/bash
# final numbers of k_run1024v3 (after partial blocks / one-block runs): SQ + traffic passes, step times
cd /root/repo
tools/profile.sh r04_cfg4shape_1024_fm_v3 --channels 1024 --frames 65536
tools/profile_lite.sh r04_1024_deno_v3 --channels 1024 --frames 65536 --demod none --no-agc-variant
STEP_M=1024 STEP_STEPS=300 python tools/step_time.py 2>&1 | grep "^region"
STEP_M=1024 STEP_DEMOD=none STEP_STEPS=300 python tools/step_time.py 2>&1 | grep "^region"
STEP_M=1024 STEP_AGC=10 STEP_STEPS=100 python tools/step_time.py 2>&1 | grep "^no timer"
