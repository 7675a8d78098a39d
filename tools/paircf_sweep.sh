mkdir -p gpurun_out/r3v
for v in default paircf default paircf; do
  if [ $v = default ]; then L=""; A=""; else L="$PWD/composable_sdr_amd/variants/libcsdr_$v.so"; A=1; fi
  echo "== $v" >> gpurun_out/r3v/pc.txt
  CSDR_LIB=$L CSDR_PAIR_ALIGN_CF=$A STEP_DEMOD=none STEP_STEPS=400 python tools/step_time.py 2>&1 | grep -E "^event pair" >> gpurun_out/r3v/pc.txt
done
cat gpurun_out/r3v/pc.txt
CSDR_LIB=$PWD/composable_sdr_amd/variants/libcsdr_paircf.so CSDR_PAIR_ALIGN_CF=1 timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "fused256 or chain_deno_matches" 2>&1 | tail -2
