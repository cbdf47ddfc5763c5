#!/bin/bash
# On the GPU box (round 6, call d): (1) streams per pass x hardware queues: CFHD_AMD_STREAMS = 1 / 2 / 3 with the runtime's 4 hardware queues and with 16;
# (2) the host-fed frame queue by batch size, passes in flight and direction; (3) the C ABI figures of this tree and of round 5's library, three runs each.
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out/r06d; O=gpurun_out/r06d
B="--steps 30 --warmup 3 --no-cpu-baseline --no-c-abi --no-other-workloads"
for q in 4 16; do for s in 1 2 3; do for d in 4; do
  GPU_MAX_HW_QUEUES=$q CFHD_AMD_STREAMS=$s python bench.py $B --depth $d > $O/b_q${q}_s${s}_d$d.json 2> $O/b_q${q}_s${s}_d$d.err
  python -c "import json;d=json.loads(open('$O/b_q${q}_s${s}_d$d.json').read().strip().splitlines()[-1]);print('queues $q streams $s depth $d fps', d['value'], 'parity', d['config']['parity_checked'])" 2>&1 | tail -1
done; done; done
for s in 1 3; do GPU_MAX_HW_QUEUES=4 CFHD_AMD_STREAMS=$s python bench.py $B --depth 3 > $O/b_q4_s${s}_d3.json 2>/dev/null; python -c "import json;d=json.loads(open('$O/b_q4_s${s}_d3.json').read().strip().splitlines()[-1]);print('queues 4 streams $s depth 3 fps', d['value'])" 2>&1 | tail -1; done
python - <<'PY' 2>&1 | tee $O/host_fed.txt
import os, sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import bench, cfhd_testlib as T
frames, pitch = T.qbist_frames(10, 8, 1920, 1080)
for q in (None,):
    for batch, depth in ((128, 4), (256, 3), (64, 6), (128, 2), (256, 4)):
        r = bench.host_fed("1080p", frames, pitch, batch=batch, depth=depth, steps=6 * depth)
        print("round trip  batch", batch, "depth", depth, {k: r.get(k) for k in ("fps", "ms_per_pass", "pcie_gbs_both_directions", "error")}, flush=True)
# encode only: frames in, samples out (no picture download)
bench.WORKLOADS["1080p-enc"] = dict(bench.WORKLOADS["1080p"], mode=1)
for batch, depth in ((128, 4), (256, 3)):
    r = bench.host_fed("1080p-enc", frames, pitch, batch=batch, depth=depth, steps=6 * depth)
    print("encode only batch", batch, "depth", depth, {k: r.get(k) for k in ("fps", "ms_per_pass", "pcie_gbs_both_directions", "error")}, flush=True)
PY
T=$(mktemp); python - <<PY
import sys; sys.path.insert(0, "tests")
import cfhd_testlib as T, numpy as np
frames, pitch = T.qbist_frames(10, 8, 1920, 1080)
open("$T", "wb").write(b"".join(f.reshape(1080, pitch)[:, :3840].tobytes() for f in frames))
PY
for lib in new r05; do for k in 1 2 3; do
  if [ $lib = r05 ]; then export LD_LIBRARY_PATH=$PWD/cineform-sdk_amd/variants/r05; else unset LD_LIBRARY_PATH; fi
  echo "c_abi $lib run $k: $(CFHD_AMD_DEVICE=0 tools/_build/cabi_bench 1920 1080 $T 8 1.5 0 8 8 | tail -1)"
done; done 2>&1 | tee $O/c_abi_ab.txt
unset LD_LIBRARY_PATH; rm -f $T
