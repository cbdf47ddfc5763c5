#!/bin/bash
# On the GPU box (round 5, call o): do the passes of the frame queue share hardware queues?  Both queue forms at three / four steps in flight with 4 (default), 8 and 16 hardware queues.
cd ${GRAFT_REPO_ROOT:-/root/repo}; R=$PWD; mkdir -p gpurun_out/r05o; O=gpurun_out/r05o
B="--steps 30 --warmup 4 --no-cpu-baseline --no-c-abi --no-other-workloads"
for hq in 4 8 16; do for q in events thread; do for d in 3 4; do
  GPU_MAX_HW_QUEUES=$hq CFHD_AMD_QUEUE=$q python bench.py $B --depth $d > $O/${q}_${d}_hq$hq.json 2> $O/${q}_${d}_hq$hq.err
  python -c "import json;d=json.loads(open('$O/${q}_${d}_hq$hq.json').read().strip().splitlines()[-1]);print('hw queues $hq queue $q depth $d', d['value'], 'fps', d['ms_per_step'], 'ms per step')"
done; done; done
