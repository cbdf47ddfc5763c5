#!/bin/bash
# On the GPU box (round 4, call p): interlaced decode with LH / HH as block lists: tests, then the 1080i line with (1) and without (0).
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
( time timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "interlaced or other_configurations" ) > gpurun_out/$1_tests.log 2>&1; tail -4 gpurun_out/$1_tests.log
for mode in 1 0 1; do
  CFHD_AMD_DEC_BLOCKS=$mode python bench.py --workload 1080i --depth 1 --steps 10 --warmup 2 --no-cpu-baseline --no-c-abi --no-other-workloads > gpurun_out/$1_d1_$mode.json 2>/dev/null
  CFHD_AMD_DEC_BLOCKS=$mode python bench.py --workload 1080i --depth 3 --steps 12 --warmup 3 --no-cpu-baseline --no-c-abi --no-other-workloads > gpurun_out/$1_d3_$mode.json 2>/dev/null
  python - $1 $mode <<'PY'
import json, sys
a = json.loads(open("gpurun_out/%s_d1_%s.json" % (sys.argv[1], sys.argv[2])).read().strip().splitlines()[-1])
b = json.loads(open("gpurun_out/%s_d3_%s.json" % (sys.argv[1], sys.argv[2])).read().strip().splitlines()[-1])
k = a["config"]["kernel_ms_per_step"]
print("lists", sys.argv[2], "depth1", a["value"], "depth3", b["value"], {n: k[n] for n in k if n.startswith("k_dec_tiles") or n.startswith("k_inv_frame")}, a["config"]["parity"]["decoded_frames_in_dither_interval"])
PY
done
