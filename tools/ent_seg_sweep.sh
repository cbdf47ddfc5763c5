#!/bin/bash
# On the GPU box: segment length of the entropy coder (coefficients per lane of k_ent_count / k_ent_emit).  usage: tools/ent_seg_sweep.sh
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for cfg in "16 256" "32 256" "32 512" "8 256"; do
  set -- $cfg
  rm -f cineform-sdk_amd/build/*.o
  make -C cineform-sdk_amd EXTRA="-DCFHD_ENT_PER_THREAD=$1 -DCFHD_ENT_TOK_CAP=$2" > /dev/null 2>&1 || { echo "build failed for $cfg"; continue; }
  for uniq in 32; do
    line=$(python bench.py --steps 10 --warmup 3 --unique $uniq --no-cpu-baseline --no-c-abi 2>/dev/null | tail -1)
    echo "per_thread=$1 tok_cap=$2 unique=$uniq $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); k=d["config"]["kernel_ms_per_step"]; print("fps", d["value"], "parity", d["config"]["parity_checked"], {n: k[n] for n in k if n.startswith("k_ent")})')"
  done
done 2>&1 | tee gpurun_out/ent_seg_sweep.log
rm -f cineform-sdk_amd/build/*.o
make -C cineform-sdk_amd > /dev/null 2>&1
