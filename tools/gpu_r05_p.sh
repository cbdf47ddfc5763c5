#!/bin/bash
# On the GPU box (round 5, call p): hardware queues and the frame queue, second look -- 16 / 24 / 32 queues, two to four steps in flight, both queue forms; the host-fed
# round trip of tools/cabi_bench with 4 and 16 queues (three runs each); the 1080i and 2160p side lines.
cd ${GRAFT_REPO_ROOT:-/root/repo}; R=$PWD; mkdir -p gpurun_out/r05p; O=gpurun_out/r05p
B="--steps 30 --warmup 4 --no-cpu-baseline --no-c-abi --no-other-workloads"
for hq in 16 24 32; do for q in events thread; do for d in 2 3; do
  GPU_MAX_HW_QUEUES=$hq CFHD_AMD_QUEUE=$q python bench.py $B --depth $d > $O/${q}_${d}_hq$hq.json 2> $O/${q}_${d}_hq$hq.err
  python -c "import json;d=json.loads(open('$O/${q}_${d}_hq$hq.json').read().strip().splitlines()[-1]);print('hw queues $hq queue $q depth $d', d['value'], 'fps', d['ms_per_step'], 'ms per step')"
done; done; done
python - <<'PY'
import sys
sys.path.insert(0, "tests")
import cfhd_testlib as T
frames, pitch = T.qbist_frames(10, 8, 1920, 1080, T.PIX_YUY2)
with open("/tmp/frames.yuy2", "wb") as f:
    for fr in frames: f.write(fr.reshape(1080, pitch)[:, :3840].tobytes())
PY
for hq in 4 16; do for r in 1 2 3; do echo "== cabi_bench 8 + 8, hardware queues $hq, run $r"; GPU_MAX_HW_QUEUES=$hq CFHD_AMD_DEVICE=0 tools/_build/cabi_bench 1920 1080 /tmp/frames.yuy2 8 1.5 0 8 8 2>&1 | tail -1; done; done
for wl in 1080i 2160p; do for q in events thread; do
  CFHD_AMD_QUEUE=$q python bench.py $B --workload $wl > $O/${wl}_$q.json 2> $O/${wl}_$q.err
  python -c "import json;d=json.loads(open('$O/${wl}_$q.json').read().strip().splitlines()[-1]);print('$wl queue $q (16 hardware queues, depth 3)', d['value'], 'fps', d['ms_per_step'], 'ms per step')"
done; done
