#!/bin/bash
# On the GPU box (round 5, call r): the host-fed many-thread case (tools/cabi_bench, 8 pool workers + 8 decoder threads) against the number of hardware queues.
cd ${GRAFT_REPO_ROOT:-/root/repo}; R=$PWD
python - <<'PY'
import sys
sys.path.insert(0, "tests")
import cfhd_testlib as T
frames, pitch = T.qbist_frames(10, 8, 1920, 1080, T.PIX_YUY2)
with open("/tmp/frames.yuy2", "wb") as f:
    for fr in frames: f.write(fr.reshape(1080, pitch)[:, :3840].tobytes())
PY
for hq in 1 2 4 8; do for r in 1 2; do echo "== cabi_bench 8 + 8, hardware queues $hq, run $r"; GPU_MAX_HW_QUEUES=$hq CFHD_AMD_DEVICE=0 tools/_build/cabi_bench 1920 1080 /tmp/frames.yuy2 8 1.5 0 8 8 2>&1 | tail -1; done; done
