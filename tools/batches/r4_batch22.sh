#!/bin/bash
# k_run1024v3 as the default FM kernel of the 1024-channel chain: the 1024-channel tests, then trace + HBM traffic passes of the cfg4 shape
cd /root/repo
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x -k "1024" 2>&1 | tail -5
tools/profile_lite.sh r04_cfg4shape_1024_fm_v3 --channels 1024 --frames 65536 --no-agc-variant
