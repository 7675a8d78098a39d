#!/bin/bash
# round 4, batch 7: checkpointed repairs: bit identity again, sweep again (the erratic segment lengths of batch 6), bench
O=gpurun_out/r4g; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q -s -k "tile_major or agc_tail or agc_whole_chunk or (run_sized and 10.0)" > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
grep -E "passed|failed|rc=|tile-major|bit-identical|mismatch|Error|error|^E |steady" $O/tests.log | tail -30
for L in 0 1056 1280 1312 1440 1568 1696 2048 2080; do
  if [ $L = 0 ]; then unset CSDR_AGC_L_TM; else export CSDR_AGC_L_TM=$L; fi
  echo "L_TM=$L: $(STEP_AGC=10 STEP_STEPS=300 timeout 300 python tools/step_time.py 2>&1 | grep -E '^(no timer)' | sed -e 's/; kernel.*//' | tr '\n' ' ')" >> $O/sweep.txt
done
unset CSDR_AGC_L_TM
cat $O/sweep.txt
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; python - <<'PY'
import json
r=json.load(open("gpurun_out/r4g/bench.json"))
print({k:(r[k] if not isinstance(r[k],dict) else {kk:r[k][kk] for kk in list(r[k])[:6]}) for k in ("value","ms_per_step","sustained_long","sustained")}, r["roofline"]["launch_ms"], r["roofline"]["frac"])
a=r["agc_variant"]; print({k:a[k] for k in ("value","ms_per_step","steps","segments_checked","segments_recomputed","hbm_roofline_frac_whole_step")})
PY
