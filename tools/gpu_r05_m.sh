#!/bin/bash
# On the GPU box (round 5, call m): the default bench line (what the driver runs), the evidence of one step at a time (kernel trace + the two PMC passes), depth and batch side lines.
cd ${GRAFT_REPO_ROOT:-/root/repo}; R=$PWD; mkdir -p gpurun_out/r05m; O=gpurun_out/r05m
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?"; python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05m/bench_default.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"], d["cpu_baseline"])
c = d["config"]
print({k: c[k] for k in c if k.startswith("kernel_ms")})
print(c.get("c_abi_fps")); print(c.get("parity")); print({k: (v.get("value"), v.get("roofline", {}).get("frac")) for k, v in (c.get("other_workloads") or {}).items()})
PY
timeout 600 bash tools/profile_round.sh r05_final 512 --depth 1 > /dev/null 2>&1; head -40 gpurun_out/prof_r05_final/summary.txt | cut -c1-200
B="--steps 20 --warmup 3 --no-cpu-baseline --no-c-abi --no-other-workloads"
for d in 1 2 3 4; do python bench.py $B --depth $d > $O/depth_$d.json 2> $O/depth_$d.err; python -c "import json;d=json.loads(open('$O/depth_$d.json').read().strip().splitlines()[-1]);print('depth $d', d['value'], d['ms_per_step'])"; done
for b in 256 1024; do python bench.py $B --batch $b > $O/batch_$b.json 2> $O/batch_$b.err; python -c "import json;d=json.loads(open('$O/batch_$b.json').read().strip().splitlines()[-1]);print('batch $b', d['value'], d['ms_per_step'])"; done
