#!/bin/bash
# On the GPU box (round 6, call i): the tile kernel's flush with all LDS reads up front (this tree) against one read per chunk (variants/h); C ABI figures, five runs per library, same box.
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out/r06i; O=gpurun_out/r06i
B="--steps 20 --warmup 3 --no-cpu-baseline --no-c-abi --no-other-workloads"
for v in new h new h; do
  lib=""; [ $v != new ] && lib=$PWD/cineform-sdk_amd/variants/$v/libcfhd_amd.so
  CFHD_AMD_LIB=$lib python bench.py $B --depth 1 > $O/t.json 2>/dev/null; python -c "import json;d=json.loads(open('$O/t.json').read().strip().splitlines()[-1]);k=d['config']['kernel_ms_per_step'];print('lib=$v depth 1 fps', d['value'], {n:k[n] for n in k if n.startswith('k_dec')})"
  CFHD_AMD_LIB=$lib python bench.py $B --depth 4 > $O/t.json 2>/dev/null; python -c "import json;d=json.loads(open('$O/t.json').read().strip().splitlines()[-1]);print('lib=$v depth 4 fps', d['value'])"
done
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
