#!/bin/bash
# On the GPU box (round 4, call i): the whole GPU suite in one piece, smoke(), the default bench line, and the rocprof evidence of the 1080p line one step at a time.
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider ) > gpurun_out/$1_tests.log 2>&1; tail -6 gpurun_out/$1_tests.log
( time python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) > gpurun_out/$1_smoke.log 2>&1; tail -2 gpurun_out/$1_smoke.log | head -1
( time python bench.py ) > gpurun_out/$1_default.json 2> gpurun_out/$1_default.err; tail -2 gpurun_out/$1_default.err; python - $1 <<'PY'
import json, sys
d = json.loads(open("gpurun_out/%s_default.json" % sys.argv[1]).read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"], d["config"].get("steps_in_flight"), json.dumps(d["config"].get("c_abi_fps"))[:700])
print(json.dumps(d["config"].get("kernel_ms_one_step_at_a_time")))
print(json.dumps({k: (v.get("value"), v.get("roofline", {}).get("kernel"), v.get("roofline", {}).get("frac")) for k, v in d["config"].get("other_workloads", {}).items()}))
print(json.dumps(d.get("cpu_baseline")))
PY
bash tools/profile_round.sh $1_1080p_depth1 512 --depth 1 > gpurun_out/$1_profile_depth1.log 2>&1; tail -2 gpurun_out/$1_profile_depth1.log
