# sustained step time and board power of the default build and of timing-only ablation builds (tools/build_variant.sh ablN kernels_fused_v2.hip -DV2_ABLATE=N)
mkdir -p gpurun_out/r3g
for v in default abl1 abl2 abl62; do
  if [ $v = default ]; then L=""; else L="$PWD/composable_sdr_amd/variants/libcsdr_$v.so"; fi
  echo "== $v" >> gpurun_out/r3g/power.txt
  CSDR_LIB=$L POWER=1 POWER_SECONDS=4 STEP_STEPS=50 python tools/step_time.py 2>&1 | grep -E "^region|smi|sustained" | sed -e "s/'Temperature[^,]*, //" -e "s/'fclk[^,]*, //g" -e "s/'mclk[^,]*, //g" -e "s/'sclk clock level:[^,]*, //" | tail -6 >> gpurun_out/r3g/power.txt
done
cat gpurun_out/r3g/power.txt
