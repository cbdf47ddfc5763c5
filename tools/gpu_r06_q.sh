#!/bin/bash
# On the GPU box (round 6, call q): the level-1 kernels with fewer instructions (dither word per block, v_sat_pk_u8_i16, 24-bit multiplies in the quantizer): kernel times one step at a time,
# four steps in flight, and the GPU tests of the 4:2:2 routes.
cd ${GRAFT_REPO_ROOT:-/root/repo}; R=$PWD; mkdir -p gpurun_out/r06q; O=gpurun_out/r06q
for d in 1 4; do timeout 600 python bench.py --depth $d --steps 40 --warmup 3 --no-cpu-baseline --no-c-abi --no-other-workloads > $O/bench_d$d.json 2> $O/bench_d$d.err; done
python - <<'PY'
import json
for d in (1, 4):
    try:
        j = json.loads(open("gpurun_out/r06q/bench_d%d.json" % d).read().strip().splitlines()[-1])
        c = j["config"]; k = c.get("kernel_ms_per_step") or c.get("kernel_ms_one_step_at_a_time")
        print("depth", d, "fps", j["value"], "parity", c.get("parity_checked"), j["roofline"]["kernel"], j["roofline"]["frac"]); print(k)
    except Exception as e:
        print("depth", d, "failed", e); print(open("gpurun_out/r06q/bench_d%d.err" % d).read()[-1500:])
PY
( time timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "block or yuy or YUY or 422 or strip or interlac or 1080 or dither or uyvy or gop or fwd or inv or quant or plane" ) > $O/tests.log 2>&1; tail -6 $O/tests.log
