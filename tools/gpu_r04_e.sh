#!/bin/bash
# On the GPU box (round 4, call e): full GPU suite (the harness test inside it runs the whole `TestCFHD -D` table), smoke(), the harness once more on its own with its
# printout kept, then the default bench line (three steps in flight) with everything in it.
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
( time CFHD_HARNESS_SECONDS=400 timeout 1700 python -m pytest tests -m gpu -x -q -p no:cacheprovider ) > gpurun_out/$1_tests.log 2>&1; tail -8 gpurun_out/$1_tests.log
( time python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) > gpurun_out/$1_smoke.log 2>&1; tail -2 gpurun_out/$1_smoke.log
( cd /tmp; time OMP_NUM_THREADS=1 timeout 900 ${GRAFT_REPO_ROOT:-/root/repo}/oracle/_ref/TestCFHD_amd -D ) > gpurun_out/$1_testcfhd_amd_D.txt 2>&1; tail -4 gpurun_out/$1_testcfhd_amd_D.txt
python - $1 <<'PY'
import json, sys
sys.path.insert(0, "tools")
from gen_testcfhd_fixture import parse_harness_output
got = parse_harness_output(open("gpurun_out/%s_testcfhd_amd_D.txt" % sys.argv[1]).read())
want = json.load(open("tests/golden/testcfhd_D.json"))["sections"]
ok = bad = unstable = 0
for k, s in enumerate(got):
    w = want[k]
    spread = max([0.1] + [max(f["psnr_seen"]) - min(f["psnr_seen"]) for f in w["frames"] if f.get("stable", True) and f["psnr_seen"]])
    for i, (size, db) in enumerate(s["frames"]):
        f = w["frames"][i]; seen = f["psnr_seen"] or [f["psnr"]]
        if not f.get("stable", True): unstable += 1; bad += size != f["size"]; continue
        good = size == f["size"] and min(seen) - spread - 0.0001 <= db <= max(seen) + spread + 0.0001
        ok += good; bad += not good
        if not good: print("MISMATCH", s["format"], s["encode"], s["decode"], i + 1, (size, db), (f["size"], seen))
print("harness: %d sections printed (%d complete), %d frames equal to the reference's printout, %d not, %d frames on which the reference has no stable PSNR (size equal)" % (len(got), sum(len(s["frames"]) == 10 for s in got), ok, bad, unstable))
PY
( time python bench.py ) > gpurun_out/$1_default.json 2> gpurun_out/$1_default.err; tail -2 gpurun_out/$1_default.err; python - $1 <<'PY'
import json, sys
d = json.loads(open("gpurun_out/%s_default.json" % sys.argv[1]).read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"], d["config"].get("steps_in_flight"), json.dumps(d["config"].get("c_abi_fps"))[:600])
print(json.dumps(d["config"].get("kernel_ms_one_step_at_a_time")))
print(json.dumps({k: (v.get("value"), v.get("roofline", {}).get("frac")) for k, v in d["config"].get("other_workloads", {}).items()}))
print(json.dumps(d.get("cpu_baseline")))
PY
bash tools/profile_round.sh $1_1080p_depth1 512 --depth 1 > gpurun_out/$1_profile_depth1.log 2>&1; tail -3 gpurun_out/$1_profile_depth1.log
