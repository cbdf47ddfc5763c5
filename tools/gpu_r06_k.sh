#!/bin/bash
# On the GPU box (round 6, call k): C ABI figures, five runs, this tree (no idle stream in pool workers) against round 5's library; host-fed queue from plain buffers; quick bench line.
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out/r06k; O=gpurun_out/r06k
T=$(mktemp); python - <<PY
import sys; sys.path.insert(0, "tests")
import cfhd_testlib as T
frames, pitch = T.qbist_frames(10, 8, 1920, 1080)
open("$T", "wb").write(b"".join(f.reshape(1080, pitch)[:, :3840].tobytes() for f in frames))
PY
for k in 1 2 3 4 5; do for lib in new r05; do
  if [ $lib = r05 ]; then export LD_LIBRARY_PATH=$PWD/cineform-sdk_amd/variants/r05; else unset LD_LIBRARY_PATH; fi
  echo "c_abi $lib run $k: $(CFHD_AMD_DEVICE=0 tools/_build/cabi_bench 1920 1080 $T 8 1.5 0 8 8 | tail -1)"
done; done 2>&1 | tee $O/c_abi_ab.txt
unset LD_LIBRARY_PATH; rm -f $T
python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "fed_from_host or pool or concurrent" 2>&1 | tail -2
python - <<'PY' 2>&1 | tee $O/host_fed_plain.txt
import sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import bench, cfhd_testlib as T
frames, pitch = T.qbist_frames(10, 8, 1920, 1080)
for reg, depth in ((False, 4), (False, 6), (True, 6)):
    r = bench.host_fed("1080p", frames, pitch, batch=128, depth=depth, steps=4 * depth, registered=reg)
    print("registered", reg, "depth", depth, {k: r.get(k) for k in ("fps", "ms_per_pass", "pcie_gbs_both_directions", "error")}, flush=True)
PY
python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-c-abi --no-other-workloads > $O/b.json 2>/dev/null; python -c "import json;d=json.loads(open('$O/b.json').read().strip().splitlines()[-1]);print('fps', d['value'], d['roofline']['kernel'], d['roofline']['frac'], d['config']['kernel_ms_one_step_at_a_time'])"
