#!/bin/bash
# On the GPU box: GPU tests that go through the batched entropy encoder, then the 1080p bench with and without the split count, and 1080i.
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
( time python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "batched or bench_size or interlaced or pool or rate_feedback" ) > gpurun_out/$1_tests.log 2>&1; tail -5 gpurun_out/$1_tests.log
bash tools/gpu_probe.sh $1 "CFHD_X=0" "CFHD_AMD_COUNT_SPLIT=0" | cut -c1-420
python bench.py --workload 1080i --steps 10 --warmup 3 --no-cpu-baseline --no-c-abi > gpurun_out/$1_1080i.json 2> gpurun_out/$1_1080i.err
python - <<PY
import json
d = json.loads(open("gpurun_out/$1_1080i.json").read().strip().splitlines()[-1])
print("1080i fps", d["value"], d["config"]["parity"])
PY
