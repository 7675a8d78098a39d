mkdir -p gpurun_out/r3h
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "submit_device or fused256 or full_size_cfg3 or strong_dc or second_generation or mask_route or bench_layout_cfg3" > gpurun_out/r3h/pytest.txt 2>&1; tail -4 gpurun_out/r3h/pytest.txt
for v in default nopair nt sc1; do
  if [ $v = default ]; then L=""; else L="$PWD/composable_sdr_amd/variants/libcsdr_$v.so"; fi
  echo "== $v" >> gpurun_out/r3h/var.txt
  CSDR_LIB=$L STEP_STEPS=400 python tools/step_time.py 2>&1 | grep -E "^region" >> gpurun_out/r3h/var.txt
done
cat gpurun_out/r3h/var.txt
CSDR_TRACE=2 python tools/trace_tiles.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3h/trace2.txt
