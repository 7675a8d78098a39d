#!/bin/bash
cd /root/repo
tools/profile_lite.sh r04_cfg4shape_1024_fm_v3 --channels 1024 --frames 65536 --no-agc-variant
tools/profile_lite.sh r04_1024_deno_v3 --channels 1024 --frames 65536 --demod none --no-agc-variant
python bench.py --channels 1024 --frames 65536 --no-agc-variant > gpurun_out/r04_bench_cfg4shape_1024ch_fm.json 2> gpurun_out/r04_bench_cfg4.err
tail -c 300 gpurun_out/r04_bench_cfg4.err
