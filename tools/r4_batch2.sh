#!/bin/bash
# round 4, batch 2: k_run256v3 parity + timing against k_run256v2
O=gpurun_out/r4b; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "run256_v3 or bench_channel_shard" > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
grep -E "passed|failed|rc=|vs v2|Error|error" $O/tests.log | tail -20
for i in 1 2; do for v in 0 1; do
  echo "v3=$v: $(CSDR_RUN_V3=$v STEP_STEPS=800 timeout 300 python tools/step_time.py 2>&1 | grep -E '^(no timer|event pair)' | sed -e 's/ per step.*kernel/ kernel/' | tr '\n' ' ')" >> $O/v3_time.txt
done; done
cat $O/v3_time.txt
for v in 0 1; do echo "== v3=$v" >> $O/power.txt; CSDR_RUN_V3=$v POWER=1 POWER_SECONDS=4 STEP_STEPS=50 timeout 300 python tools/step_time.py 2>&1 | grep -E "smi|sustained" | sed -e "s/'Temperature[^,]*, //" -e "s/'fclk[^,]*, //g" -e "s/'mclk[^,]*, //g" -e "s/'sclk clock level:[^,]*, //" | tail -6 >> $O/power.txt; done
cat $O/power.txt
CSDR_RUN_V3=1 STEP_DEMOD=none STEP_STEPS=400 timeout 300 python tools/step_time.py 2>&1 | grep -E '^(no timer|event pair)' > $O/v3_deno.txt; CSDR_RUN_V3=0 STEP_DEMOD=none STEP_STEPS=400 timeout 300 python tools/step_time.py 2>&1 | grep -E '^(no timer|event pair)' >> $O/v3_deno.txt; cat $O/v3_deno.txt
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; python - <<'PY'
import json
r=json.load(open("gpurun_out/r4b/bench.json"))
print({k:(r[k] if not isinstance(r[k],dict) else {kk:r[k][kk] for kk in list(r[k])[:6]}) for k in ("value","ms_per_step","cold_window","sustained_long","sustained")}, r["roofline"]["launch_ms"], r["roofline"]["frac"])
PY
