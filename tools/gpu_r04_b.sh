#!/bin/bash
# On the GPU box (round 4, call b): the 1080p bench line with the level-1 block lists (default) and without (CFHD_AMD_BLOCKS=0), forced-strip + bench-size GPU tests.
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
( time timeout 600 python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "strip_kernels or bench_sizes" ) > gpurun_out/$1_tests.log 2>&1; tail -4 gpurun_out/$1_tests.log
B="--steps 10 --warmup 2 --no-cpu-baseline --no-c-abi --no-other-workloads"
python bench.py $B > gpurun_out/$1_blocks.json 2> gpurun_out/$1_blocks.err; tail -2 gpurun_out/$1_blocks.err
CFHD_AMD_BLOCKS=0 python bench.py $B > gpurun_out/$1_dense.json 2> gpurun_out/$1_dense.err
CFHD_AMD_DENSE_L1=1 python bench.py $B > gpurun_out/$1_both.json 2> gpurun_out/$1_both.err
CFHD_AMD_COUNT_SPLIT=0 python bench.py $B > gpurun_out/$1_nosplit.json 2> gpurun_out/$1_nosplit.err
python - $1 <<'PY'
import json,sys
for tag in ("blocks","dense","both","nosplit"):
    try:
        d=json.loads(open("gpurun_out/%s_%s.json"%(sys.argv[1],tag)).read().strip().splitlines()[-1])
        print(tag, d["value"], d["ms_per_step"], json.dumps(d["config"].get("kernel_ms_per_step")))
    except Exception as e: print(tag, "failed", e)
PY
