#!/bin/bash
# Builds cineform-sdk_amd/variants/<tag>/libcfhd_amd.so: the whole library with extra compiler flags for both .hip files (A/B runs: CFHD_AMD_LIB=... python bench.py).
# usage: tools/build_variant_all.sh <tag> <flags...>
set -e
cd "$(dirname "$0")/../cineform-sdk_amd"
TAG=$1; shift
mkdir -p variants/$TAG build
make -s > /dev/null
for f in cfhd_entropy_gpu cfhd_device; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-result -Icsrc -I../include "$@" --offload-arch=gfx950 -c csrc/$f.hip -o build/variant_${TAG}_$f.o
done
OBJ=$(ls build/*.o | grep -v "variant_\|\.hip\.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -Wl,-Bsymbolic -Wl,--version-script=exports.map -o variants/$TAG/libcfhd_amd.so $OBJ build/variant_${TAG}_cfhd_entropy_gpu.o build/variant_${TAG}_cfhd_device.o -lpthread
echo "built variants/$TAG/libcfhd_amd.so ($*)"
