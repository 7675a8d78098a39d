#!/bin/bash
# Builds composable_sdr_amd/variants/libcsdr_run256v3.so: the product library + round 4's k_run256v3 (one 512-thread workgroup per CU,
# front / back wave roles; measured NOT faster than k_run256v2: DESIGN.md 4.1d, profiles/r04_run256v3_one_wg_per_cu.txt).  The kernel
# lives here, outside the product (verdict r04 #8).  Use: CSDR_LIB=.../libcsdr_run256v3.so CSDR_DIAG=1 CSDR_RUN_V3=1 python tools/step_time.py
set -e
cd "$(dirname "$0")/../../composable_sdr_amd/csrc"
make -s -j8 > /dev/null
mkdir -p ../variants build/var_run256v3
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -I. -DCSDR_WITH_RUN256_V3=1"
/opt/rocm/bin/hipcc $F -c kernels_fused.hip -o build/var_run256v3/kernels_fused.hip.o
/opt/rocm/bin/hipcc $F -c ../../tools/variants/kernels_run256_v3.hip -o build/var_run256v3/kernels_run256_v3.hip.o
OBJS=$(ls build/*.o | grep -v "build/kernels_fused.hip.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../variants/libcsdr_run256v3.so $OBJS build/var_run256v3/kernels_fused.hip.o build/var_run256v3/kernels_run256_v3.hip.o -ldl
echo "built variants/libcsdr_run256v3.so"
