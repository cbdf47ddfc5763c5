#!/bin/bash
# Builds cineform-sdk_amd/variants/libcfhd_amd_<tag>.so: the library with the entropy driver (cfhd_entropy_gpu.hip: every k_ent_* / k_dec_* kernel) compiled with extra
# flags, for A/B runs on the GPU box (CFHD_AMD_LIB=... python bench.py).  usage: tools/build_variant.sh <tag> <flags...>
set -e
cd "$(dirname "$0")/../cineform-sdk_amd"
TAG=$1; shift
mkdir -p variants build
make -s > /dev/null
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-result -Icsrc -I../include "$@" --offload-arch=gfx950 -c csrc/cfhd_entropy_gpu.hip -o build/variant_$TAG.o
OBJ=$(ls build/*.o | grep -v "variant_\|cfhd_entropy_gpu.hip.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -Wl,-Bsymbolic -Wl,--version-script=exports.map -o variants/libcfhd_amd_$TAG.so $OBJ build/variant_$TAG.o -lpthread
echo "built variants/libcfhd_amd_$TAG.so ($*)"
