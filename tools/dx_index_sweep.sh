#!/bin/bash
# On the GPU box: rebuild the entropy driver with different register budgets of k_dec_index (memo entries per lane, waves per SIMD asked of the
# compiler, workgroups per CU in the grid) and time the bench's decoder kernels.  usage: tools/dx_index_sweep.sh  (gpurun_out/dx_index_sweep.log)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for cfg in "3 5 5" "2 5 5" "6 5 5" "3 5 6" "2 6 6"; do
  set -- $cfg
  attr=""; [ "$2" != "0" ] && attr="-DCFHD_DX_WAVES=$2"
  rm -f cineform-sdk_amd/build/cfhd_entropy_gpu.hip.o
  make -C cineform-sdk_amd EXTRA="-DCFHD_DX_MEMO=$1 $attr" > /dev/null 2>&1 || { echo "build failed for $cfg"; continue; }
  for uniq in 8 32; do
    line=$(CFHD_AMD_DX_GRID_INDEX=$((256 * $3)) python bench.py --steps 10 --warmup 3 --unique $uniq --no-cpu-baseline --no-c-abi 2>/dev/null | tail -1)
    echo "memo=$1 waves_per_eu=$2 wg_per_cu=$3 unique=$uniq $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); k=d["config"]["kernel_ms_per_step"]; print("fps", d["value"], "parity", d["config"]["parity_checked"], {n: k[n] for n in k if n.startswith("k_dec")})')"
  done
done 2>&1 | tee gpurun_out/dx_index_sweep.log
rm -f cineform-sdk_amd/build/cfhd_entropy_gpu.hip.o
make -C cineform-sdk_amd > /dev/null 2>&1
