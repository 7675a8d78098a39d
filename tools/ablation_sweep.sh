mkdir -p gpurun_out/r3f
for v in default abl1 abl2 abl4 abl8 abl16 abl32 abl62; do
  if [ $v = default ]; then L=""; else L="$PWD/composable_sdr_amd/variants/libcsdr_$v.so"; fi
  echo "== $v" >> gpurun_out/r3f/abl.txt
  CSDR_LIB=$L STEP_STEPS=400 python tools/step_time.py 2>&1 | grep -E "^region|^no timer" >> gpurun_out/r3f/abl.txt
done
echo "== default, zero input" >> gpurun_out/r3f/abl.txt
ZERO_INPUT=1 python tools/kernel_time.py fm 2>&1 | tail -1 >> gpurun_out/r3f/abl.txt
python tools/kernel_time.py fm 2>&1 | tail -1 >> gpurun_out/r3f/abl.txt
cat gpurun_out/r3f/abl.txt
