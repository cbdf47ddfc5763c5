#!/bin/bash
# On the GPU box (round 5, call s): with 16 hardware queues the streams inside a pass really run beside each other -- do the two overlaps inside a pass (level-1 count beside
# the level-2 / level-3 transforms, parser beside k_ent_emit) still pay, one and three steps in flight?  And the turns per stage of the frame queue.
cd ${GRAFT_REPO_ROOT:-/root/repo}; R=$PWD; mkdir -p gpurun_out/r05s; O=gpurun_out/r05s
B="--steps 30 --warmup 4 --no-cpu-baseline --no-c-abi --no-other-workloads"
run() { tag=$1; shift; for d in 1 3; do env "$@" python bench.py $B --depth $d > $O/${tag}_$d.json 2> $O/${tag}_$d.err; python -c "import json;d=json.loads(open('$O/${tag}_$d.json').read().strip().splitlines()[-1]);print('$tag depth $d', d['value'], 'fps', d['ms_per_step'], 'ms per step')"; done; }
run default A=1
run count_on_main_stream CFHD_AMD_COUNT_SPLIT=0
run parser_behind_payloads CFHD_AMD_PARSE_EARLY=0
run both_off CFHD_AMD_COUNT_SPLIT=0 CFHD_AMD_PARSE_EARLY=0
run unordered CFHD_AMD_QUEUE=unordered
run both_off_unordered CFHD_AMD_COUNT_SPLIT=0 CFHD_AMD_PARSE_EARLY=0 CFHD_AMD_QUEUE=unordered
