#!/bin/bash
# The decode routes added at the end of round 3 and the ones whose output stages were restated, on the hardware: tools/gpu_call_10.sh <tag>
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
( time timeout 250 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "bayer_decode or rgba4444_decode or half_resolution or yu64_decode or rgb8_decode or rgb10_decode or rgba8_encode or b64a_decode_equals or rgb444_decode_to_b64a or rg30 or yuv422_decode_to_rg24 or encode_matches_golden" ) > gpurun_out/$1_tests.log 2>&1; tail -5 gpurun_out/$1_tests.log
( time timeout 40 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) > gpurun_out/$1_smoke.log 2>&1; tail -2 gpurun_out/$1_smoke.log
