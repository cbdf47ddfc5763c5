#!/bin/bash
# On the GPU box (round 5, call u): the encode-only lines (byr4-2160p, rg48-2160p: every step ends in a host copy of 130-270 MB of samples) against steps in flight and the queue's form.
cd ${GRAFT_REPO_ROOT:-/root/repo}; R=$PWD; mkdir -p gpurun_out/r05u; O=gpurun_out/r05u
B="--steps 30 --warmup 4 --no-cpu-baseline --no-c-abi --no-other-workloads"
for wl in byr4-2160p rg48-2160p; do for q in default ordered thread; do for d in 1 2 3; do
  CFHD_AMD_QUEUE=$q python bench.py $B --workload $wl --depth $d > $O/${wl}_${q}_$d.json 2> $O/${wl}_${q}_$d.err
  python -c "import json;d=json.loads(open('$O/${wl}_${q}_$d.json').read().strip().splitlines()[-1]);print('$wl queue $q depth $d', d['value'], 'fps', d['ms_per_step'], 'ms per step')"
done; done; done
