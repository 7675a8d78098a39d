#!/bin/bash
cd /root/repo
tools/profile.sh r04_cfg2_m64_deno --channels 64 --frames 1048576 --demod none
tools/profile.sh r04_cfg3_deno --demod none
tools/profile.sh r04_cfg4shape_1024_fm_v3 --channels 1024 --frames 65536
tools/profile.sh r04_1024_deno_v3 --channels 1024 --frames 65536 --demod none
