#!/bin/bash
# On the GPU box (round 5, call a): the single-pass entropy decoder (CFHD_AMD_DEC=emit: k_dec_index_emit + k_dec_scatter) against the two-pass one -- a slice of
# the GPU suite on the new arrangement, then the 1080p line of both at depth 1 and 3, then a kernel trace of the new arrangement.
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out/r05a; O=gpurun_out/r05a
( time CFHD_AMD_DEC=emit timeout 600 python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "batched or decode_reference or round_trip or fuzz or concurrent or interlaced_decode" ) > $O/tests_emit.log 2>&1; tail -4 $O/tests_emit.log
B="python bench.py --no-other-workloads --no-c-abi --no-cpu-baseline"
for dec in tiles emit tiles emit; do
  for d in 1 3; do
    CFHD_AMD_DEC=$dec $B --depth $d --steps 20 --warmup 3 > $O/${dec}_d$d.json 2> $O/${dec}_d$d.err
    python - $O/${dec}_d$d.json $dec $d <<'PY'
import json, sys
try:
    a = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    k = a["config"].get("kernel_ms_one_step_at_a_time") or a["config"]["kernel_ms_per_step"]
    print(sys.argv[2], "depth", sys.argv[3], a["value"], "fps", {n: round(v, 3) for n, v in k.items() if n.startswith("k_dec") or n.startswith("k_inv_yuv")}, a["config"]["parity_checked"])
except Exception as e:
    print(sys.argv[2], sys.argv[3], "failed", e)
PY
  done
done
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
CFHD_AMD_DEC=emit rocprofv3 --kernel-trace --stats -d $R/$O/trace -o bench -- python $R/bench.py --depth 1 --steps 5 --warmup 2 --no-cpu-baseline --no-c-abi --no-other-workloads > $R/$O/emit_traced.json 2> $R/$O/trace.err
T=$(find $R/$O/trace -name '*.db' | head -1)
python $R/tools/rocprof_summary.py $T > $R/$O/emit_trace_summary.txt 2> $R/$O/summary.err
rm -rf $R/$O/trace
head -30 $R/$O/emit_trace_summary.txt
