#!/bin/bash
# round 5, batch 3: full GPU suite after the library cut (k_run256v3 / k_dc_pick_tile out, CSDR_DIAG gating), threaded synth; smoke; bench default
O=gpurun_out/r5c; mkdir -p $O
python -m pytest tests -m gpu -q --durations=25 > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
tail -40 $O/tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
python bench.py --shard channel --mix --channels 4096 --frames 16384 --demod none --steps 5 --warmup 1 --no-cpu-baseline --preheat-ms 300 > $O/bench_mix_c.json 2> $O/bench_mix_c.err; wc -l $O/bench_mix_c.json
