#!/bin/bash
# On the GPU box (round 6, call r): k_ent_emit as resident waves (k_ent_emit_stream) against a wave per segment: kernel times one step at a time for several grids, the step rate with
# four steps in flight, and the GPU tests of the encode side.
cd ${GRAFT_REPO_ROOT:-/root/repo}; R=$PWD; mkdir -p gpurun_out/r06r; O=gpurun_out/r06r
B="--steps 30 --warmup 3 --no-cpu-baseline --no-c-abi --no-other-workloads"
run() { # name, depth, env...
	local n=$1 d=$2; shift 2
	env "$@" timeout 600 python bench.py $B --depth $d > $O/bench_$n.json 2> $O/bench_$n.err
	python - "$n" <<'PY'
import json, sys
n = sys.argv[1]
try:
    j = json.loads(open("gpurun_out/r06r/bench_%s.json" % n).read().strip().splitlines()[-1])
    c = j["config"]; k = c.get("kernel_ms_per_step") or c.get("kernel_ms_one_step_at_a_time")
    print(n, "fps", j["value"], "parity", c.get("parity_checked"), "emit", k.get("k_ent_emit"), "roofline", j["roofline"]["kernel"], j["roofline"]["frac"])
except Exception as e:
    print(n, "failed", e); print(open("gpurun_out/r06r/bench_%s.err" % n).read()[-1200:])
PY
}
run plain_d1 1 CFHD_AMD_EMIT=plain
for g in 1024 2048 3072 4096; do run stream${g}_d1 1 CFHD_AMD_EMIT=stream CFHD_AMD_EMIT_GRID=$g; done
run plain_d4 4 CFHD_AMD_EMIT=plain
run stream2048_d4 4 CFHD_AMD_EMIT=stream CFHD_AMD_EMIT_GRID=2048
run stream4096_d4 4 CFHD_AMD_EMIT=stream CFHD_AMD_EMIT_GRID=4096
( time timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "sample or encode or entropy or 1080 or 2160 or queue" ) > $O/tests.log 2>&1; tail -5 $O/tests.log
