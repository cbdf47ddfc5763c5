#!/bin/bash
# On the GPU box (round 5, call q): the frame queue's forms with 16 hardware queues -- launcher thread (default) | launches on the caller's thread (inline) | the blocking
# pass of round 4 on a thread (thread) -- on the 1080p, 2160p and 1080i lines.
cd ${GRAFT_REPO_ROOT:-/root/repo}; R=$PWD; mkdir -p gpurun_out/r05q; O=gpurun_out/r05q
B="--steps 30 --warmup 4 --no-cpu-baseline --no-c-abi --no-other-workloads"
for wl in 1080p 2160p 1080i; do for q in launcher inline thread; do for d in 2 3; do
  CFHD_AMD_QUEUE=$q python bench.py $B --workload $wl --depth $d > $O/${wl}_${q}_$d.json 2> $O/${wl}_${q}_$d.err
  python -c "import json;d=json.loads(open('$O/${wl}_${q}_$d.json').read().strip().splitlines()[-1]);print('$wl queue $q depth $d', d['value'], 'fps', d['ms_per_step'], 'ms per step')"
done; done; done
