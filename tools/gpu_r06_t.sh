#!/bin/bash
# On the GPU box (round 6, call t): kernel times one step at a time and the step rate with four in flight -- for A/B builds (occupancy attributes of the strip kernels).
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out/r06t; O=gpurun_out/r06t
for d in 1 4; do timeout 600 python bench.py --depth $d --steps 40 --warmup 3 --no-cpu-baseline --no-c-abi --no-other-workloads > $O/bench_d$d.json 2> $O/bench_d$d.err; done
python - <<'PY'
import json
for d in (1, 4):
    try:
        j = json.loads(open("gpurun_out/r06t/bench_d%d.json" % d).read().strip().splitlines()[-1])
        c = j["config"]; k = c.get("kernel_ms_per_step") or c.get("kernel_ms_one_step_at_a_time")
        print("depth", d, "fps", j["value"], "parity", c.get("parity_checked"), j["roofline"]["kernel"], j["roofline"]["frac"]); print(k)
    except Exception as e:
        print("depth", d, "failed", e); print(open("gpurun_out/r06t/bench_d%d.err" % d).read()[-1500:])
PY
