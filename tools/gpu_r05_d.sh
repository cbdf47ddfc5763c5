#!/bin/bash
# On the GPU box (round 5, call d): what bounds the host-fed round trip through the C ABI (tools/cabi_bench: pool of 8 workers + 8 decoder threads)?  The same binary under
# different runtime settings: hardware queues per process, SDMA on / off, direct dispatch.
cd ${GRAFT_REPO_ROOT:-/root/repo}; R=$PWD; mkdir -p gpurun_out/r05d; O=gpurun_out/r05d
python - <<'PY'
import sys, os
sys.path.insert(0, "tests")
import cfhd_testlib as T
frames, pitch = T.qbist_frames(10, 8, 1920, 1080, T.PIX_YUY2)
with open("/tmp/frames.yuy2", "wb") as f:
    for fr in frames: f.write(fr.reshape(1080, pitch)[:, :3840].tobytes())
PY
run() { echo "== $*"; env "$@" CFHD_AMD_DEVICE=0 tools/_build/cabi_bench 1920 1080 /tmp/frames.yuy2 8 1.5 0 8 8 2>&1 | tail -1; }
run HSA_ENABLE_SDMA=1
run HSA_ENABLE_SDMA=1
run HSA_ENABLE_SDMA=0
run HSA_ENABLE_SDMA=1 GPU_MAX_HW_QUEUES=2
run HSA_ENABLE_SDMA=1 GPU_MAX_HW_QUEUES=8
run HSA_ENABLE_SDMA=1 GPU_MAX_HW_QUEUES=16
run HSA_ENABLE_SDMA=1 GPU_MAX_HW_QUEUES=24
run HSA_ENABLE_SDMA=1 AMD_DIRECT_DISPATCH=0
run HSA_ENABLE_SDMA=1 HIP_FORCE_DEV_KERNARG=1
run HSA_ENABLE_SDMA=1 CFHD_AMD_ENCODE_BATCH=8
run HSA_ENABLE_SDMA=1 CFHD_AMD_ENCODE_BATCH=8 GPU_MAX_HW_QUEUES=8
