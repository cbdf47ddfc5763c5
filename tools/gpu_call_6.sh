#!/bin/bash
# On the GPU box: Bayer tests, then the byr4-2160p workload with the fused level 1 and with the component planes (CFHD_AMD_BAYER=planes).
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
( time python -m pytest tests -m gpu -q -p no:cacheprovider -k "byr" ) > gpurun_out/$1_tests.log 2>&1; tail -5 gpurun_out/$1_tests.log
for E in "CFHD_X=0"; do
  env $E python bench.py --workload byr4-2160p --steps 10 --warmup 3 --no-cpu-baseline --no-c-abi > gpurun_out/$1_byr4.json 2> gpurun_out/$1_byr4.err
  python - "$E" <<PY
import json, sys
try:
    d = json.loads(open("gpurun_out/$1_byr4.json").read().strip().splitlines()[-1])
    print(sys.argv[1], "fps", d["value"], d["config"]["kernel_ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"], d["config"].get("parity"))
except Exception as e:
    print(sys.argv[1], "failed:", e, open("gpurun_out/$1_byr4.err").read()[-400:])
PY
done 2>&1 | tee gpurun_out/$1_byr4.log
