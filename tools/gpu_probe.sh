#!/bin/bash
# On the GPU box: the bench under a list of environment settings ("NAME=VALUE" each), kernel times per setting.  tools/gpu_probe.sh <tag> "<env1>" "<env2>" ...
TAG=$1; shift
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
for E in "$@"; do
  env $E python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-c-abi --no-other-workloads > gpurun_out/${TAG}_probe.json 2> gpurun_out/${TAG}_probe.err
  python - "$E" <<PY
import json, sys
try:
    d = json.loads(open("gpurun_out/${TAG}_probe.json").read().strip().splitlines()[-1])
    k = d["config"]["kernel_ms_per_step"]
    print(sys.argv[1], "fps", d["value"], {n: k[n] for n in k if n.startswith("k_ent") or n.startswith("k_dec")}, d["config"].get("dx_stats"), d["config"].get("parity"))
except Exception as e:
    print(sys.argv[1], "failed:", e, open("gpurun_out/${TAG}_probe.err").read()[-400:])
PY
done 2>&1 | tee gpurun_out/${TAG}_probe.log
