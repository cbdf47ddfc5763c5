#!/bin/bash
# Every GPU test that touches a decode route added or changed at the end of round 3 (half resolution of all formats, Bayer -> BYR4, the restated 8- / 10-bit output stages): tools/gpu_call_11.sh <tag>
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
( time timeout 230 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "half_resolution or bayer_decode or rgba4444_decode or yu64_decode or v210_decode or rgb8_decode or rgb10_decode or rgba8_encode or b64a_decode_equals or yuv422_decode_to_rg24 or invalid_arguments or decode_reference_samples or gop" ) > gpurun_out/$1_tests.log 2>&1; tail -5 gpurun_out/$1_tests.log
