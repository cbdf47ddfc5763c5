#!/bin/bash
# On the GPU box: strip vs tiled transform kernels at small batch sizes (what the synchronous C ABI and the gathered passes launch)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for n in 1 2 4 8 16 32 64; do
  for mode in strip tile; do
    if [ $mode = tile ]; then export CFHD_AMD_PLANES=tile CFHD_AMD_FORWARD=tile CFHD_AMD_INVERSE=tile; else unset CFHD_AMD_PLANES CFHD_AMD_FORWARD CFHD_AMD_INVERSE; fi
    python bench.py --batch $n --unique $n --steps 30 --warmup 3 --no-cpu-baseline --no-c-abi 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin); k=d['config']['kernel_ms_per_step']
print('n=$n $mode fps=%.0f' % d['value'], ' '.join('%s=%.3f' % (a.replace('k_','').replace('_strip',''), b) for a,b in k.items() if 'fwd' in a or 'inv' in a))"
  done
done
