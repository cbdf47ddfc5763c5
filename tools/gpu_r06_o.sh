#!/bin/bash
# On the GPU box (round 6, call o): k_ent_emit with all its loads in front of its first branch (this tree) against the tree before (variants/k2), and the whole library compiled
# with kernel arguments preloaded into SGPRs (variants/preload: -mllvm -amdgpu-kernarg-preload-count=16).
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out/r06o; O=gpurun_out/r06o
B="--steps 20 --warmup 3 --no-cpu-baseline --no-c-abi --no-other-workloads"
for v in k2 new preload k2 new preload; do
  lib=""; [ $v != new ] && lib=$PWD/cineform-sdk_amd/variants/$v/libcfhd_amd.so
  CFHD_AMD_LIB=$lib python bench.py $B --depth 1 > $O/t.json 2>$O/t.err; python -c "import json;d=json.loads(open('$O/t.json').read().strip().splitlines()[-1]);k=d['config']['kernel_ms_per_step'];print('lib=$v depth 1 fps', d['value'], 'parity', d['config']['parity_checked'], {n.split('[')[0]:round(k[n],3) for n in k})" 2>&1 | tail -1
  CFHD_AMD_LIB=$lib python bench.py $B --depth 4 > $O/t.json 2>/dev/null; python -c "import json;d=json.loads(open('$O/t.json').read().strip().splitlines()[-1]);print('lib=$v depth 4 fps', d['value'], 'parity', d['config']['parity_checked'])" 2>&1 | tail -1
done
