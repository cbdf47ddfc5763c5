#!/bin/bash
# On the GPU box (round 4, call j): the C ABI from plain host buffers (tools/cabi_bench.cpp) with the staging copy in 1 / 2 / 4 / 8 pieces.
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
python - <<'PY'
import sys
sys.path.insert(0, "tests")
from cfhd_testlib import synth_yuy2
with open("/tmp/frames.yuy2", "wb") as f:
    for i in range(8): f.write(synth_yuy2(1920, 1080, 20 + i)[0].tobytes())
PY
export CFHD_AMD_DEVICE=0
for p in 1 4 1 4; do
  for t in 8 16; do
    echo "pieces $p threads $t: $(CFHD_AMD_STAGE_PIECES=$p tools/_build/cabi_bench 1920 1080 /tmp/frames.yuy2 8 1.5 0 $t $t | tail -1)"
  done
done 2>&1 | tee gpurun_out/$1_cabi_pieces.txt
