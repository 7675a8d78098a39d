#!/bin/bash
O=gpurun_out/r4i; mkdir -p $O
export TMPDIR=/tmp
for v in default tmabl5; do
  if [ $v = default ]; then LIB=""; else LIB="$PWD/composable_sdr_amd/variants/libcsdr_$v.so"; fi
  for L in 1568 3104; do
  CSDR_LIB=$LIB CSDR_AGC_L_TM=$L rocprofv3 --kernel-trace --stats -f csv -d $O/t_${v}_$L -o t -- python3 tools/kernel_time.py fm 256 262144 10 > $O/t_${v}_$L.log 2>&1
  echo "== $v L=$L"; python3 - $O/t_${v}_$L <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "csdr" in r["Name"] and ("spec" in r["Name"] or "fix" in r["Name"] or "run256" in r["Name"]): print(f"  {r['Name'][40:90]:50s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.1f} us")
PY
done; done
find $O -name "*.csv" -size +2M -delete; find $O -type f -name "*.db" -delete
