#!/bin/bash
# On the GPU box (round 6, call x): kernel trace of the C ABI bench with one decoder handle and one pool worker -- what a single-frame decode / encode spends where.
cd ${GRAFT_REPO_ROOT:-/root/repo}; R=$PWD; mkdir -p gpurun_out/r06x; O=$R/gpurun_out/r06x
python - <<'PY'
import sys, os
sys.path.insert(0, "tests")
import cfhd_testlib as T
frames, pitch = T.qbist_frames(10, 8, 1920, 1080)
with open("/tmp/frames.yuy2", "wb") as f:
    for fr in frames: f.write(fr.reshape(1080, pitch)[:, :3840].tobytes())
PY
cd /tmp && export TMPDIR=/tmp
CFHD_AMD_DEVICE=0 $R/tools/_build/cabi_bench 1920 1080 /tmp/frames.yuy2 8 2 0 1 1 | tail -1 | cut -c1-400
CFHD_AMD_DEVICE=0 timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace -o cabi -- $R/tools/_build/cabi_bench 1920 1080 /tmp/frames.yuy2 8 2 0 1 1 > $O/traced.json 2> $O/trace.err
T=$(find $O/trace -name '*.db' | head -1)
python $R/tools/rocprof_summary.py $T > $O/summary.txt 2>&1; head -45 $O/summary.txt | cut -c1-150
rm -rf $O/trace
