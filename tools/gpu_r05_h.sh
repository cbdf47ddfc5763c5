#!/bin/bash
# On the GPU box (round 5, call h): interlaced two-frame groups on the hardware (+ the rest of the group tests and the pool / harness-protocol tests), then the
# synchronous C-ABI rates with the round-1 band decoder for a single frame (CFHD_AMD_DEC=par) against the chunk-indexed one.
cd ${GRAFT_REPO_ROOT:-/root/repo}; R=$PWD; mkdir -p gpurun_out/r05h; O=gpurun_out/r05h
( time timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "gop or pool or harness_calls or interlaced" ) > $O/tests.log 2>&1; tail -6 $O/tests.log
python - <<'PY'
import sys
sys.path.insert(0, "tests")
import cfhd_testlib as T
frames, pitch = T.qbist_frames(10, 8, 1920, 1080, T.PIX_YUY2)
with open("/tmp/frames.yuy2", "wb") as f:
    for fr in frames: f.write(fr.reshape(1080, pitch)[:, :3840].tobytes())
PY
run() { echo "== $*"; env "$@" CFHD_AMD_DEVICE=0 tools/_build/cabi_bench 1920 1080 /tmp/frames.yuy2 8 1.0 0 8 8 2>&1 | tail -1; }
run A=1
run CFHD_AMD_DEC=par
run A=2
run CFHD_AMD_DEC=par
run CFHD_AMD_DECODE_BATCH=0
