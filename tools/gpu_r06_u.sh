#!/bin/bash
# On the GPU box (round 6, call u: the final tree of the round): the evidence of the tree -- one step at a time: kernel trace + the two PMC passes + SQ counters; the default bench line; the whole GPU suite.
cd ${GRAFT_REPO_ROOT:-/root/repo}; R=$PWD; mkdir -p gpurun_out/r06u; O=gpurun_out/r06u
timeout 900 bash tools/profile_round.sh r06_final4 512 --depth 1 > /dev/null 2>&1; head -60 gpurun_out/prof_r06_final4/summary.txt | cut -c1-220
timeout 600 bash tools/profile_sq.sh r06_sq4 --depth 1 > /dev/null 2>&1
for p in 0 1 2; do grep "k_dec_tiles\|k_dec_index\|k_ent_emit\|k_ent_count_blocks\|k_inv_yuv422_strip_blocks\|k_fwd_yuv422_strip_blocks" gpurun_out/sq_r06_sq4/summary_p$p.txt | grep "SQ_" | sed 's/_ZN4cfhd3dev[0-9]*\(k_[a-z0-9_]*\)[A-Za-z0-9_]*/\1/' | awk '{printf "%-28s %-24s %s\n", $1, $2, $5}'; done | tee $O/sq_short.txt | head -80
( time timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | grep real; echo "bench done"; python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r06u/bench_default.json").read().strip().splitlines()[-1])
    print(d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"].get("traffic"), d.get("cpu_baseline", {}).get("value"))
    print("host_fed", {k: (v if not isinstance(v, dict) else {a: v.get(a) for a in ("fps", "ms_per_pass", "pcie_gbs_both_directions", "error")}) for k, v in d.get("host_fed", {}).items()})
    c = d["config"]; print(c.get("with_16_hardware_queues")); print({k: c[k] for k in c if k.startswith("kernel_ms_one")}); print("parity", c.get("parity"))
    print({k: (v.get("value"), v.get("roofline", {}).get("frac"), v.get("error")) for k, v in (c.get("other_workloads") or {}).items()})
    print("c_abi", {k: (v if not isinstance(v, dict) else {a: b for a, b in v.items() if a != "runs"}) for k, v in (c.get("c_abi_fps") or {}).items()})
except Exception as e:
    print("failed", e); print(open("gpurun_out/r06u/bench_default.err").read()[-2000:])
PY
( time timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider ) > $O/tests.log 2>&1; tail -6 $O/tests.log
