#!/bin/bash
# On the GPU box (round 4, call a): full GPU suite, smoke(), then the 1080p bench line with the level-1 block lists (default) and without (CFHD_AMD_BLOCKS=0).
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider ) > gpurun_out/$1_tests.log 2>&1; tail -12 gpurun_out/$1_tests.log
( time python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) > gpurun_out/$1_smoke.log 2>&1; tail -2 gpurun_out/$1_smoke.log
B="--steps 10 --warmup 2 --no-cpu-baseline --no-c-abi --no-other-workloads"
python bench.py $B > gpurun_out/$1_blocks.json 2> gpurun_out/$1_blocks.err; tail -2 gpurun_out/$1_blocks.err
CFHD_AMD_BLOCKS=0 python bench.py $B > gpurun_out/$1_dense.json 2> gpurun_out/$1_dense.err
CFHD_AMD_DENSE_L1=1 python bench.py $B > gpurun_out/$1_both.json 2> gpurun_out/$1_both.err
python - <<'PY'
import json,sys
for tag in ("blocks","dense","both"):
    try:
        d=json.loads(open("gpurun_out/%s_%s.json"%("r04a",tag)).read().strip().splitlines()[-1])
        print(tag, d["value"], d["ms_per_step"], json.dumps(d["config"].get("kernel_ms_per_step")) )
    except Exception as e: print(tag, "failed", e)
PY
