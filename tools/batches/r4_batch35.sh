#!/bin/bash
# tile-major plane for the AGC tail behind k_run1024v3<CF32>: bit identity, the 1024-channel tests, step time against the row-major plane
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -s -k "tile_major_route_at_1024 or pfb1024 or run1024_v3" 2>&1 | tail -6
for tm in 1 0; do
  echo -n "CSDR_AGC_TM=$tm: "; CSDR_AGC_TM=$tm STEP_M=1024 STEP_AGC=10 STEP_STEPS=100 timeout 300 python tools/step_time.py 2>&1 | grep -E "^no timer" | cut -c27-
done
