#!/bin/bash
# On the GPU box (round 4, call d): full GPU suite, smoke(), then the reference's own harness linked against the library run to its end (every section the library
# serves: all twenty rows at full resolution, the first five at half) and compared with the committed fixture.
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider ) > gpurun_out/$1_tests.log 2>&1; tail -8 gpurun_out/$1_tests.log
( time python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) > gpurun_out/$1_smoke.log 2>&1; tail -2 gpurun_out/$1_smoke.log
( cd /tmp; time OMP_NUM_THREADS=1 timeout 900 ${GRAFT_REPO_ROOT:-/root/repo}/oracle/_ref/TestCFHD_amd -D ) > gpurun_out/$1_testcfhd_amd_D.txt 2>&1
python - $1 <<'PY'
import json, sys
sys.path.insert(0, "tools")
from gen_testcfhd_fixture import parse_harness_output
got = parse_harness_output(open("gpurun_out/%s_testcfhd_amd_D.txt" % sys.argv[1]).read())
want = json.load(open("tests/golden/testcfhd_D.json"))["sections"]
ok = bad = 0
for k, s in enumerate(got):
    w = want[k]
    for i, (size, db) in enumerate(s["frames"]):
        f = w["frames"][i]
        seen = f["psnr_seen"] or [f["psnr"]]
        good = size == f["size"] and (not f.get("stable", True) or min(seen) - 0.1001 <= db <= max(seen) + 0.1001)
        ok += good; bad += not good
        if not good: print("MISMATCH", s["format"], s["encode"], s["decode"], i + 1, (size, db), (f["size"], seen))
print("harness: %d sections printed (%d complete), %d frames equal to the reference's printout, %d not" % (len(got), sum(len(s["frames"]) == 10 for s in got), ok, bad))
PY
