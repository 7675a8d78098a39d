#!/bin/bash
# Builds composable_sdr_amd/variants/libcsdr_<NAME>.so with extra compiler flags for ONE translation unit
# (timing experiments: load it with CSDR_LIB).  Usage: tools/build_variant.sh NAME FILE.hip -DFOO=1 ...
set -e
NAME=$1; SRC=$2; shift 2
cd "$(dirname "$0")/../composable_sdr_amd/csrc"
make -s -j8 > /dev/null
mkdir -p ../variants build/var_$NAME
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -Wno-invalid-offsetof "$@" -c $SRC -o build/var_$NAME/$SRC.o
OBJS=$(ls build/*.o | grep -v "build/$SRC.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../variants/libcsdr_$NAME.so $OBJS build/var_$NAME/$SRC.o -ldl
echo "built variants/libcsdr_$NAME.so"
