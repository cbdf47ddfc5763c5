#!/bin/bash
# On the GPU box (round 5, call t): free-running passes (the default again), 16 hardware queues: steps in flight 2..6 on the 1080p line; the side lines at 3; 8 and 32 hardware queues at 3.
cd ${GRAFT_REPO_ROOT:-/root/repo}; R=$PWD; mkdir -p gpurun_out/r05t; O=gpurun_out/r05t
B="--steps 30 --warmup 4 --no-cpu-baseline --no-c-abi --no-other-workloads"
for d in 2 3 4 5 6; do python bench.py $B --depth $d > $O/depth_$d.json 2> $O/depth_$d.err; python -c "import json;d=json.loads(open('$O/depth_$d.json').read().strip().splitlines()[-1]);print('1080p depth $d', d['value'], 'fps', d['ms_per_step'], 'ms per step')"; done
for hq in 8 32; do GPU_MAX_HW_QUEUES=$hq python bench.py $B --depth 3 > $O/hq_$hq.json 2> $O/hq_$hq.err; python -c "import json;d=json.loads(open('$O/hq_$hq.json').read().strip().splitlines()[-1]);print('1080p depth 3 hardware queues $hq', d['value'], 'fps', d['ms_per_step'], 'ms per step')"; done
for wl in 1080i 2160p byr4-2160p rg48-2160p b64a-4320p; do python bench.py $B --workload $wl > $O/$wl.json 2> $O/$wl.err; python -c "import json;d=json.loads(open('$O/$wl.json').read().strip().splitlines()[-1]);print('$wl depth 3', d['value'], 'fps', d['ms_per_step'], 'ms per step', d['config']['parity'].get('samples_equal_reference_encoder'))"; done
