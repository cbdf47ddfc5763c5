#!/bin/bash
# On the GPU box (round 6, call v): the default bench line of the final tree (python bench.py, no flags), timed.
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out/r06v; O=gpurun_out/r06v
( time timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | grep real
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06v/bench_default.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["launch_ms"])
print({k: (v.get("value"), v.get("roofline", {}).get("kernel"), v.get("roofline", {}).get("frac")) for k, v in d["config"]["other_workloads"].items()})
PY
