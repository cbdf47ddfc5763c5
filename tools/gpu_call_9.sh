#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
( time timeout 240 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "batched or bench_size or interlaced_samples or concurrent" ) > gpurun_out/$1_tests.log 2>&1; tail -4 gpurun_out/$1_tests.log
timeout 200 bash tools/gpu_probe.sh $1 "CFHD_X=0" | cut -c1-620
timeout 120 python bench.py --workload 1080i --steps 10 --warmup 3 --no-cpu-baseline --no-c-abi 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('1080i', d['value'], d['config']['parity'])"
