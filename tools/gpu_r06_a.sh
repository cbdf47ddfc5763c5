#!/bin/bash
# On the GPU box (round 6, call a): k_dec_tiles as a workgroup per tile against the round-5 library (a wave per tile of 4096), one step at a time and four in flight;
# tile / workgroup shapes B (28672, 512 threads), C (7168, 256), D (15360, 512); decoder tests first.
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out/r06a; O=gpurun_out/r06a
[ -n "$SKIP_TESTS" ] || ( time timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "batched or fuzz or decode_equals or interlaced or smoke or damaged or concurrent" ) > $O/tests.log 2>&1; tail -3 $O/tests.log
B="--steps 20 --warmup 3 --no-cpu-baseline --no-c-abi --no-other-workloads"
show() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1]); c = d["config"]
    k = c.get("kernel_ms_one_step_at_a_time") or c["kernel_ms_per_step"]
    print(sys.argv[1], "fps", d["value"], "parity", c["parity_checked"], {n: k[n] for n in k if n.startswith("k_dec")}, "roofline", d["roofline"]["kernel"], d["roofline"]["frac"])
except Exception as e:
    print(sys.argv[1], "failed:", e)
PY
}
for v in ${VARIANTS:-r05 "" B C D}; do
  lib=""; [ -n "$v" ] && lib=$PWD/cineform-sdk_amd/variants/libcfhd_amd_$v.so
  for d in 1 4; do
    CFHD_AMD_LIB=$lib python bench.py $B --depth $d > $O/bench_${v:-new}_d$d.json 2> $O/bench_${v:-new}_d$d.err; show "lib=${v:-new} depth=$d" $O/bench_${v:-new}_d$d.json
  done
done
