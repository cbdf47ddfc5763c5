#!/bin/bash
# On the GPU box (round 5, call w): the batched-path tests on the final tree, then the default bench line (what the driver runs).
cd ${GRAFT_REPO_ROOT:-/root/repo}; R=$PWD; mkdir -p gpurun_out/r05w; O=gpurun_out/r05w
timeout 600 python -m pytest tests -m gpu -x -q -k "frame_queue or batched or peak or bench_sizes or strip_kernels" > $O/batched_tests.log 2>&1; echo "tests rc $?"; tail -3 $O/batched_tests.log
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?"; python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05w/bench_default.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["steps"], d["roofline"]["frac"], d["cpu_baseline"])
c = d["config"]
print(c.get("c_abi_fps", {}).get("plain_buffers")); print(c.get("parity")); print({k: (v.get("value"), v.get("roofline", {}).get("frac")) for k, v in (c.get("other_workloads") or {}).items()})
PY
