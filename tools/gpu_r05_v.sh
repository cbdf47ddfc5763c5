#!/bin/bash
# On the GPU box (round 5, call v): turns for the encode halves only (default) | for both halves (ordered) | none (free), 16 hardware queues.
cd ${GRAFT_REPO_ROOT:-/root/repo}; R=$PWD; mkdir -p gpurun_out/r05v; O=gpurun_out/r05v
B="--steps 30 --warmup 4 --no-cpu-baseline --no-c-abi --no-other-workloads"
for spec in "1080p 3" "1080p 4" "byr4-2160p 3" "2160p 3" "1080i 3"; do set -- $spec; for q in default ordered free; do
  CFHD_AMD_QUEUE=$q python bench.py $B --workload $1 --depth $2 > $O/$1_${q}_$2.json 2> $O/$1_${q}_$2.err
  python -c "import json;d=json.loads(open('$O/$1_${q}_$2.json').read().strip().splitlines()[-1]);print('$1 queue $q depth $2', d['value'], 'fps', d['ms_per_step'], 'ms per step')"
done; done
