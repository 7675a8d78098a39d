mkdir -p gpurun_out/r3c
for cfg in "6 1" "6 0" "3 1" "0 1"; do
  set -- $cfg
  echo "=== WU=$1 ROT=$2" >> gpurun_out/r3c/sweep.txt
  CSDR_WU=$1 CSDR_WU_ROT=$2 STEP_STEPS=60 python tools/step_time.py 2>&1 | grep -v amdgpu.ids | tail -1 >> gpurun_out/r3c/sweep.txt
  CSDR_WU=$1 CSDR_WU_ROT=$2 CSDR_TRACE=2 python tools/trace_tiles.py 2>&1 | grep -v amdgpu.ids | head -7 >> gpurun_out/r3c/sweep.txt
done
cat gpurun_out/r3c/sweep.txt
