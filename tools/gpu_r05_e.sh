#!/bin/bash
# On the GPU box (round 5, call e): single-pass decoder, fourth build (coalesced step logs [piece][step][lane], tile pass with its fetches three stages ahead) against the
# two-pass one; then what bounds the host-fed round trip through the C ABI (tools/gpu_r05_d.sh: the same binary under different runtime settings).
cd ${GRAFT_REPO_ROOT:-/root/repo}; R=$PWD; mkdir -p gpurun_out/r05e; O=gpurun_out/r05e
( time CFHD_AMD_DEC=emit timeout 600 python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "batched or decode_reference or round_trip or fuzz or concurrent or interlaced_decode or frame_queue" ) > $O/tests_emit.log 2>&1; tail -4 $O/tests_emit.log
B="python bench.py --no-other-workloads --no-c-abi --no-cpu-baseline"
show() { python - "$@" <<'PY'
import json, sys
try:
    a = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    k = a["config"].get("kernel_ms_one_step_at_a_time") or a["config"]["kernel_ms_per_step"]
    print(" ".join(sys.argv[2:]), a["value"], "fps", {n: round(v, 3) for n, v in k.items() if n.startswith("k_dec") or n.startswith("k_inv_yuv")}, a["config"]["parity_checked"])
except Exception as e:
    print(" ".join(sys.argv[2:]), "failed", e)
PY
}
for dec in tiles emit; do for d in 1 3; do
  CFHD_AMD_DEC=$dec $B --depth $d --steps 20 --warmup 3 > $O/${dec}_d$d.json 2> $O/${dec}_d$d.err; show $O/${dec}_d$d.json $dec depth $d
done; done
for extra in "-DCFHD_DX_SC_THREADS=128" "-DCFHD_DX_SC_THREADS=512"; do
  rm -f cineform-sdk_amd/build/cfhd_entropy_gpu.hip.o
  make -C cineform-sdk_amd EXTRA="$extra" > /dev/null 2>&1 || { echo "build failed: $extra"; continue; }
  tag=$(echo "$extra" | tr -d ' =-' )
  CFHD_AMD_DEC=emit $B --depth 1 --steps 10 --warmup 3 > $O/var_$tag.json 2> $O/var_$tag.err; show $O/var_$tag.json emit depth 1 "$extra"
done
rm -f cineform-sdk_amd/build/cfhd_entropy_gpu.hip.o; make -C cineform-sdk_amd > /dev/null 2>&1
( cd /tmp && export TMPDIR=/tmp
CFHD_AMD_DEC=emit rocprofv3 --kernel-trace --stats -d $R/$O/trace -o bench -- python $R/bench.py --depth 1 --steps 5 --warmup 2 --no-cpu-baseline --no-c-abi --no-other-workloads > $R/$O/emit_traced.json 2> $R/$O/trace.err
T=$(find $R/$O/trace -name '*.db' | head -1)
python $R/tools/rocprof_summary.py $T > $R/$O/emit_trace_summary.txt 2> $R/$O/summary.err
rm -rf $R/$O/trace
head -12 $R/$O/emit_trace_summary.txt )
bash tools/gpu_r05_d.sh 2>&1 | tee $O/cabi_env_sweep.txt
