mkdir -p gpurun_out/r3i
for w in "1.2,0.8" "1.24,0.76" "1.28,0.72" "1.32,0.68"; do
  echo "== weights $w" >> gpurun_out/r3i/w.txt
  CSDR_RUN_WEIGHTS=$w STEP_STEPS=400 python tools/step_time.py 2>&1 | grep -E "^region" >> gpurun_out/r3i/w.txt
done
cat gpurun_out/r3i/w.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "submit_device or bench_layout_cfg2 or bench_layout_cfg4 or bench_layout_cfg5" 2>&1 | tail -5
