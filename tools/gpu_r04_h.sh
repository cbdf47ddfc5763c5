#!/bin/bash
# On the GPU box (round 4, call h): the interlaced / Bayer / packed tests with the new strip kernels, then the bench lines of 1080i, byr4-2160p and b64a-4320p and
# the rocprof evidence of the 1080i line.
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
( time timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "interlaced or bayer or byr4 or other_configurations or packed16 or 8k" ) > gpurun_out/$1_tests.log 2>&1; tail -4 gpurun_out/$1_tests.log
bash tools/profile_round.sh $1_1080i 512 --workload 1080i --depth 1 > gpurun_out/$1_profile_1080i.log 2>&1
for wl in 1080i byr4-2160p b64a-4320p; do
  python bench.py --workload $wl --no-cpu-baseline --no-c-abi --no-other-workloads > gpurun_out/$1_bench_$wl.json 2> gpurun_out/$1_bench_$wl.err
  python - $1 $wl <<'PY'
import json, sys
d = json.loads(open("gpurun_out/%s_bench_%s.json" % (sys.argv[1], sys.argv[2])).read().strip().splitlines()[-1])
print(sys.argv[2], d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"], json.dumps(d["config"]["kernel_ms_per_step"]))
PY
done
