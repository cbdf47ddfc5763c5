#!/bin/bash
# On the GPU box: rebuild the entropy driver with different workgroup shapes of k_dec_tiles (coefficients per tile, threads, workgroups per CU in the
# grid) and time the bench's decoder kernels.  usage: tools/dx_tile_sweep.sh  (writes gpurun_out/dx_tile_sweep.log)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
IFS=";" read -ra LIST <<< "${CFGS:-2048 512 2;2048 640 2;2048 384 3;2048 576 2}"; unset IFS
for cfg in "${LIST[@]}"; do
  set -- $cfg
  rm -f cineform-sdk_amd/build/cfhd_entropy_gpu.hip.o
  make -C cineform-sdk_amd EXTRA="-DCFHD_DX_TILE=$1 -DCFHD_DX_TILE_THREADS=$2" > /dev/null 2>&1 || { echo "build failed for $cfg"; continue; }
  for uniq in 8 32; do
    line=$(CFHD_AMD_DX_GRID_TILES=$((256 * $3)) python bench.py --steps 10 --warmup 3 --unique $uniq --no-cpu-baseline --no-c-abi 2>/dev/null | tail -1)
    echo "tile=$1 threads=$2 wg_per_cu=$3 unique=$uniq $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); k=d["config"]["kernel_ms_per_step"]; print("fps", d["value"], "parity", d["config"]["parity_checked"], {n: k[n] for n in k if n.startswith("k_dec")})')"
  done
done 2>&1 | tee gpurun_out/dx_tile_sweep.log
rm -f cineform-sdk_amd/build/cfhd_entropy_gpu.hip.o
make -C cineform-sdk_amd > /dev/null 2>&1
