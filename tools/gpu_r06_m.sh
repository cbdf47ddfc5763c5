#!/bin/bash
# On the GPU box (round 6, call m): is the crash of cabi_bench under rocprofv3 (call l) the library's or the tool's?  Three traced runs per library; kernel times of the C ABI case.
cd ${GRAFT_REPO_ROOT:-/root/repo}; R=$PWD; mkdir -p gpurun_out/r06m; O=$R/gpurun_out/r06m
T=$(mktemp); python - <<PY
import sys; sys.path.insert(0, "tests")
import cfhd_testlib as T
frames, pitch = T.qbist_frames(10, 8, 1920, 1080)
open("$T", "wb").write(b"".join(f.reshape(1080, pitch)[:, :3840].tobytes() for f in frames))
PY
cd /tmp && export TMPDIR=/tmp
for k in 1 2 3; do for lib in new r05; do
  if [ $lib = r05 ]; then export LD_LIBRARY_PATH=$R/cineform-sdk_amd/variants/r05; else unset LD_LIBRARY_PATH; fi
  rm -rf $O/trace_$lib
  rocprofv3 --kernel-trace --stats -d $O/trace_$lib -o cabi -- env CFHD_AMD_DEVICE=0 $R/tools/_build/cabi_bench 1920 1080 $T 8 0.7 0 8 8 > $O/cabi_${lib}_$k.txt 2> $O/cabi_${lib}_$k.err
  echo "traced $lib run $k: rc $? $(tail -1 $O/cabi_${lib}_$k.txt | cut -c1-200)"
  if [ $k = 1 ]; then D=$(find $O/trace_$lib -name '*.db' 2>/dev/null | head -1); [ -n "$D" ] && python - "$D" "$lib" <<'PY'
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1])
try:
    rows = db.execute("select name, count(*), avg(end-start)/1e3, sum(end-start)/1e6 from kernels group by name order by 4 desc limit 22").fetchall()
    tot = db.execute("select sum(end-start)/1e6 from kernels").fetchone()[0]
    print("lib", sys.argv[2], "kernel time total ms", round(tot, 1))
    for n, c, a, s in rows:
        m = re.search(r"k_[a-z0-9_]+", n); print("  %-34s calls %6d  avg %8.1f us  total %8.1f ms" % (m.group(0) if m else n[:34], c, a, s))
except Exception as e: print("no kernel table:", e)
PY
  fi
  rm -rf $O/trace_$lib
done; done 2>&1 | tee $O/traced.txt
grep -l "SIGSEGV" $O/*.err | head; for f in $(grep -l "SIGSEGV" $O/*.err | head -2); do grep -A24 "SIGSEGV" $f | cut -c1-200; done
rm -f $T
