#!/bin/bash
# On the GPU box (round 4, call c): the frame queue -- 1, 2, 3, 4 steps in flight (bench.py --depth), 512 and 256 frames per step.
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
B="--steps 12 --warmup 2 --no-cpu-baseline --no-c-abi --no-other-workloads"
for d in 1 2 3 4; do python bench.py $B --depth $d > gpurun_out/$1_depth$d.json 2> gpurun_out/$1_depth$d.err; tail -1 gpurun_out/$1_depth$d.err; done
for d in 2 4; do python bench.py $B --depth $d --batch 256 > gpurun_out/$1_b256_depth$d.json 2> gpurun_out/$1_b256_depth$d.err; done
python - $1 <<'PY'
import json,sys
for tag in ("depth1","depth2","depth3","depth4","b256_depth2","b256_depth4"):
    try:
        d=json.loads(open("gpurun_out/%s_%s.json"%(sys.argv[1],tag)).read().strip().splitlines()[-1])
        print(tag, d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"], d["config"]["parity"].get("frames_checked"), json.dumps(d["config"].get("kernel_ms_per_step")))
    except Exception as e: print(tag, "failed", e)
PY
