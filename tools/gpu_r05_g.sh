#!/bin/bash
# On the GPU box (round 5, call g): the reference's own harness, -E (EncodeSpeedTest: pool of 16, 500 frames per format row), linked against this library and against the
# reference; the host-fed round trip of tools/cabi_bench three times with the library's new defaults (pool workers gather while decoders are at work, kernel arguments in
# device memory), and once each with the gathering off / forced.
cd ${GRAFT_REPO_ROOT:-/root/repo}; R=$PWD; mkdir -p gpurun_out/r05g; O=gpurun_out/r05g
( cd /tmp && timeout 900 $R/oracle/_ref/TestCFHD_amd -E > $R/$O/testcfhd_amd_E.txt 2>&1; echo "TestCFHD_amd -E rc $?"; grep -a "fps\|Pixel format\|Encode:" $R/$O/testcfhd_amd_E.txt | paste - - - | head -30 )
( cd /tmp && timeout 600 $R/oracle/_ref/TestCFHD_ref -E > $R/$O/testcfhd_ref_E.txt 2>&1; echo "TestCFHD_ref -E rc $?"; grep -a "fps\|Pixel format\|Encode:" $R/$O/testcfhd_ref_E.txt | paste - - - | head -30 )
python - <<'PY'
import sys
sys.path.insert(0, "tests")
import cfhd_testlib as T
frames, pitch = T.qbist_frames(10, 8, 1920, 1080, T.PIX_YUY2)
with open("/tmp/frames.yuy2", "wb") as f:
    for fr in frames: f.write(fr.reshape(1080, pitch)[:, :3840].tobytes())
PY
run() { echo "== $*"; env "$@" CFHD_AMD_DEVICE=0 tools/_build/cabi_bench 1920 1080 /tmp/frames.yuy2 8 1.5 0 8 8 2>&1 | tail -1; }
run A=1; run A=2; run A=3
run CFHD_AMD_ENCODE_BATCH=0
run CFHD_AMD_ENCODE_BATCH=8
run HIP_FORCE_DEV_KERNARG=0
run HIP_FORCE_DEV_KERNARG=0 CFHD_AMD_ENCODE_BATCH=0
echo "== 16 + 16 threads"; CFHD_AMD_DEVICE=0 tools/_build/cabi_bench 1920 1080 /tmp/frames.yuy2 8 1.5 0 16 16 2>&1 | tail -1
echo "== registered buffers 16 + 16"; CFHD_AMD_DEVICE=0 tools/_build/cabi_bench 1920 1080 /tmp/frames.yuy2 8 1.5 1 16 16 2>&1 | tail -1
