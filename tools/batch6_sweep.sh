mkdir -p gpurun_out/r3s
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "fused256 or full_size_cfg3 or strong_dc or second_generation or submit_device or bench_layout_cfg3 or fused_interleaved" > gpurun_out/r3s/pytest.txt 2>&1; tail -3 gpurun_out/r3s/pytest.txt
for b in 1 0 1 0; do
  echo "== batch6=$b" >> gpurun_out/r3s/b6.txt
  CSDR_WU_BATCH6=$b STEP_STEPS=400 python tools/step_time.py 2>&1 | grep -E "^event pair" >> gpurun_out/r3s/b6.txt
done
cat gpurun_out/r3s/b6.txt
CSDR_TRACE=2 python tools/trace_tiles.py 2>&1 | grep -v amdgpu.ids | head -8
