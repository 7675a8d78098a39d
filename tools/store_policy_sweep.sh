# output-store cache policy of k_run256v2 (whole-line stores): default / nt / sc1 / sc0 sc1, FM and CF32, 400 steps each
mkdir -p gpurun_out/r3x
for d in fm none; do for v in default stnt stsc1 stsc0sc1 default stnt; do
  if [ $v = default ]; then L=""; else L="$PWD/composable_sdr_amd/variants/libcsdr_$v.so"; fi
  echo "== $d $v: $(CSDR_LIB=$L STEP_DEMOD=$d STEP_STEPS=400 python tools/step_time.py 2>&1 | grep -E '^event pair')" | tee -a gpurun_out/r3x/store_policy.txt
done; done
