#!/bin/bash
# On the GPU box (round 6, call e): the host link (tools/pcie_duplex), stream creation order and depth with the runtime's 4 hardware queues.
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out/r06e; O=gpurun_out/r06e
tools/_build/pcie_duplex 2>&1 | tee $O/pcie_duplex.txt
HSA_ENABLE_SDMA=0 tools/_build/pcie_duplex 2>&1 | sed 's/^/HSA_ENABLE_SDMA=0  /' | tee -a $O/pcie_duplex.txt
B="--steps 30 --warmup 3 --no-cpu-baseline --no-c-abi --no-other-workloads"
run() { env "$@" python bench.py $B --depth $D > $O/t.json 2> $O/t.err; python -c "import json;d=json.loads(open('$O/t.json').read().strip().splitlines()[-1]);print('$*', 'depth $D fps', d['value'], 'parity', d['config']['parity_checked'])" 2>&1 | tail -1; }
for D in 4 5 6; do run X=1; run CFHD_AMD_STREAM_ORDER=alt; done
D=4; run CFHD_AMD_STREAMS=3; run GPU_MAX_HW_QUEUES=16; run GPU_MAX_HW_QUEUES=8; run GPU_MAX_HW_QUEUES=8 CFHD_AMD_STREAMS=2
