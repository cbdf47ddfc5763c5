#!/bin/bash
# On the GPU box (round 6, call c): tile shapes I (8192 / 512 threads, four workgroups per CU), J (12288 / 512), K (14848 / 640) against the default (14848 / 512);
# the whole GPU suite (reference legs strict by default); the default bench line with the host-fed frame queue.
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out/r06c; O=gpurun_out/r06c
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
for v in "" I J K; do
  lib=""; [ -n "$v" ] && lib=$PWD/cineform-sdk_amd/variants/libcfhd_amd_$v.so
  for d in 1 4; do
    CFHD_AMD_LIB=$lib python bench.py $B --depth $d > $O/bench_${v:-new}_d$d.json 2> $O/bench_${v:-new}_d$d.err; show "lib=${v:-new} depth=$d" $O/bench_${v:-new}_d$d.json
  done
done
( time timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider ) > $O/tests.log 2>&1; tail -6 $O/tests.log
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?"; python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r06c/bench_default.json").read().strip().splitlines()[-1])
    print(d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"], d.get("cpu_baseline", {}).get("value"))
    print("host_fed", d.get("host_fed"))
    c = d["config"]; print({k: c[k] for k in c if k.startswith("kernel_ms")}); print("parity", c.get("parity"))
    print({k: (v.get("value"), v.get("roofline", {}).get("frac"), v.get("error")) for k, v in (c.get("other_workloads") or {}).items()})
    print("c_abi", {k: (v if not isinstance(v, dict) else {a: b for a, b in v.items() if a != "runs"}) for k, v in (c.get("c_abi_fps") or {}).items()})
except Exception as e:
    print("failed", e); print(open("gpurun_out/r06c/bench_default.err").read()[-2000:])
PY
