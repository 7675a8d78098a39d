for i in 1 2; do for v in default abl64 abl128 abl192; do
  if [ $v = default ]; then L=""; else L="$PWD/composable_sdr_amd/variants/libcsdr_$v.so"; fi
  echo "$v: $(CSDR_LIB=$L STEP_STEPS=400 python tools/step_time.py 2>&1 | grep -E '^event pair' | sed -e 's/.*kernel/kernel/')"
done; done
