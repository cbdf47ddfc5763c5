#!/bin/bash
# On the GPU box: a selection of the GPU tests, then the default bench without the CPU baseline / C-ABI legs; prints the kernel times.
#   tools/gpu_quick.sh <tag> "<pytest -k expression or empty>" [bench.py arguments]
TAG=$1; KEXPR=$2; shift 2
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
if [ -n "$KEXPR" ]; then
  ( time python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "$KEXPR" ) > gpurun_out/${TAG}_tests.log 2>&1
  tail -4 gpurun_out/${TAG}_tests.log
fi
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-c-abi --no-other-workloads "$@" > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/${TAG}_bench.json").read().strip().splitlines()[-1])
    print("fps", d["value"], "ms/step", d["ms_per_step"], "parity", d["config"]["parity"])
    print(d["config"]["kernel_ms_per_step"])
    print("roofline", d["roofline"]["kernel"], d["roofline"]["frac"])
except Exception as e:
    print("bench failed:", e); print(open("gpurun_out/${TAG}_bench.err").read()[-1500:])
PY
