#!/bin/bash
# the bench lines README / DESIGN quote (one box): default (cfg3 FM + agc_variant + cpu_baseline), DeNo, cfg2, cfg4 shape, cfg5 shape
O=gpurun_out/r4q; mkdir -p $O
python bench.py > $O/r04_bench_default.json 2> $O/err_default.txt
python bench.py --demod none --no-cpu-baseline --no-agc-variant > $O/r04_bench_deno.json 2> $O/err_deno.txt
python bench.py --channels 64 --frames 1048576 --demod none --no-cpu-baseline --no-agc-variant > $O/r04_bench_cfg2_m64_deno.json 2>> $O/err_deno.txt
python bench.py --channels 1024 --frames 65536 --no-cpu-baseline --no-agc-variant > $O/r04_bench_cfg4shape_1024ch_fm.json 2>> $O/err_deno.txt
python bench.py --channels 4096 --frames 16384 --demod none --mix --no-cpu-baseline --no-agc-variant > $O/r04_bench_cfg5shape_4096ch_mix.json 2>> $O/err_deno.txt
python bench.py --frames 524288 --no-cpu-baseline --no-agc-variant > $O/r04_bench_secondary_524288_frames.json 2>> $O/err_deno.txt
for f in $O/r04_bench_*.json; do python - $f <<'PY'
import json,sys
r=json.load(open(sys.argv[1]))
print(sys.argv[1].split('/')[-1], r["value"], r["ms_per_step"], r["roofline"]["frac"], r["roofline"]["launch_ms"], r.get("sustained_long",{}).get("ms_per_step"), r.get("sustained",{}).get("ms_per_step"), r.get("agc_variant",{}).get("ms_per_step"))
PY
done
