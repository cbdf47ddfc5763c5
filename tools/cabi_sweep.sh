#!/bin/bash
# On the GPU box: C-ABI rates for several (decoder threads, pool workers) pairs.  usage: [REG=1] [CFGS="8 8;16 16"] tools/cabi_sweep.sh
cd ${GRAFT_REPO_ROOT:-/root/repo}
python - <<'PY'
import sys, os
sys.path.insert(0, "tests")
import cfhd_testlib as T
frames, pitch = T.qbist_frames(10, 8, 1920, 1080)
with open("/tmp/frames.yuy2", "wb") as f:
    for fr in frames: f.write(fr.tobytes())
PY
IFS=';' read -ra LIST <<< "${CFGS:-8 4;8 8;16 8;16 16;32 16}"
for cfg in "${LIST[@]}"; do
  set -- $cfg
  echo "decoders=$1 workers=$2: $(tools/_build/cabi_bench 1920 1080 /tmp/frames.yuy2 8 1.5 ${REG:-0} $1 $2)"
done
