#!/bin/bash
# On the GPU box (round 5, call f): what does a wave of the tile pass wait for?  k_dec_tiles without its global stores (a probe build: no valid output, parity check off).
cd ${GRAFT_REPO_ROOT:-/root/repo}; R=$PWD; mkdir -p gpurun_out/r05f; O=gpurun_out/r05f
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
$B --depth 1 --steps 10 --warmup 3 > $O/base.json 2> $O/base.err; show $O/base.json tiles as built
rm -f cineform-sdk_amd/build/cfhd_entropy_gpu.hip.o
make -C cineform-sdk_amd EXTRA="-DCFHD_DX_PROBE_NOSTORE" > /dev/null 2>&1 || echo "build failed"
CFHD_BENCH_NO_PARITY=1 $B --depth 1 --steps 10 --warmup 3 > $O/nostore.json 2> $O/nostore.err; show $O/nostore.json tiles without stores
CFHD_BENCH_NO_PARITY=1 CFHD_AMD_DEC_BLOCKS=0 $B --depth 1 --steps 10 --warmup 3 > $O/nostore_dense.json 2> $O/nostore_dense.err; show $O/nostore_dense.json tiles without stores, dense form
rm -f cineform-sdk_amd/build/cfhd_entropy_gpu.hip.o; make -C cineform-sdk_amd > /dev/null 2>&1
CFHD_AMD_DEC_BLOCKS=0 $B --depth 1 --steps 10 --warmup 3 > $O/dense.json 2> $O/dense.err; show $O/dense.json tiles as built, dense form
( cd /tmp && timeout 600 $R/oracle/_ref/TestCFHD_amd -E > $R/$O/testcfhd_amd_E.txt 2>&1; echo "TestCFHD_amd -E rc $?"; grep -c "fps" $R/$O/testcfhd_amd_E.txt; grep "fps\|Pixel format\|Encode:" $R/$O/testcfhd_amd_E.txt | head -70 )
