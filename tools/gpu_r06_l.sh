#!/bin/bash
# On the GPU box (round 6, call l): kernel times of the C ABI case (tools/cabi_bench: single frames and gathered passes of up to eight), this tree against round 5's library.
cd ${GRAFT_REPO_ROOT:-/root/repo}; R=$PWD; mkdir -p gpurun_out/r06l; O=$R/gpurun_out/r06l
T=$(mktemp); python - <<PY
import sys; sys.path.insert(0, "tests")
import cfhd_testlib as T
frames, pitch = T.qbist_frames(10, 8, 1920, 1080)
open("$T", "wb").write(b"".join(f.reshape(1080, pitch)[:, :3840].tobytes() for f in frames))
PY
cd /tmp && export TMPDIR=/tmp
for lib in new r05; do
  if [ $lib = r05 ]; then export LD_LIBRARY_PATH=$R/cineform-sdk_amd/variants/r05; else unset LD_LIBRARY_PATH; fi
  rocprofv3 --kernel-trace --stats -d $O/trace_$lib -o cabi -- env CFHD_AMD_DEVICE=0 $R/tools/_build/cabi_bench 1920 1080 $T 8 0.7 0 8 8 > $O/cabi_$lib.txt 2> $O/cabi_$lib.err
  tail -1 $O/cabi_$lib.txt
  D=$(find $O/trace_$lib -name '*.db' | head -1)
  python - "$D" "$lib" <<'PY'
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, count(*), avg(end-start)/1e3, sum(end-start)/1e6 from kernels group by name order by 4 desc limit 24").fetchall()
tot = db.execute("select sum(end-start)/1e6 from kernels").fetchone()[0]
print("lib", sys.argv[2], "kernel time total ms", round(tot, 1))
for n, c, a, s in rows:
    m = re.search(r"k_[a-z0-9_]+", n); print("  %-34s calls %6d  avg %8.1f us  total %8.1f ms" % (m.group(0) if m else n[:34], c, a, s))
PY
  rm -rf $O/trace_$lib
done 2>&1 | tee $O/kernels.txt
unset LD_LIBRARY_PATH
for k in 1 2 3; do for lib in new r05; do
  if [ $lib = r05 ]; then export LD_LIBRARY_PATH=$R/cineform-sdk_amd/variants/r05; else unset LD_LIBRARY_PATH; fi
  echo "c_abi $lib run $k: $(CFHD_AMD_DEVICE=0 $R/tools/_build/cabi_bench 1920 1080 $T 8 1.5 0 8 8 | tail -1)"
done; done 2>&1 | tee $O/c_abi_ab.txt
rm -f $T
