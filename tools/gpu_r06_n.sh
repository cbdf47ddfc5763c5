#!/bin/bash
# On the GPU box (round 6, call n): kernel times of the C ABI case with this tree's library (rocprofv3 crashes in about half of its runs of this many-thread program with either library: retried).
cd ${GRAFT_REPO_ROOT:-/root/repo}; R=$PWD; mkdir -p gpurun_out/r06n; O=$R/gpurun_out/r06n
T=$(mktemp); python - <<PY
import sys; sys.path.insert(0, "tests")
import cfhd_testlib as T
frames, pitch = T.qbist_frames(10, 8, 1920, 1080)
open("$T", "wb").write(b"".join(f.reshape(1080, pitch)[:, :3840].tobytes() for f in frames))
PY
cd /tmp && export TMPDIR=/tmp
for k in 1 2 3 4 5 6; do
  rm -rf $O/trace
  rocprofv3 --kernel-trace --stats -d $O/trace -o cabi -- env CFHD_AMD_DEVICE=0 $R/tools/_build/cabi_bench 1920 1080 $T 8 0.7 0 8 8 > $O/cabi_$k.txt 2> $O/cabi_$k.err
  rc=$?; echo "traced new run $k: rc $rc $(tail -1 $O/cabi_$k.txt | cut -c1-200)"
  if [ $rc = 0 ]; then D=$(find $O/trace -name '*.db' | head -1); python - "$D" <<'PY'
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, count(*), avg(end-start)/1e3, sum(end-start)/1e6 from kernels group by name order by 4 desc limit 22").fetchall()
tot = db.execute("select sum(end-start)/1e6 from kernels").fetchone()[0]
print("lib new kernel time total ms", round(tot, 1))
for n, c, a, s in rows:
    m = re.search(r"k_[a-z0-9_]+", n); print("  %-34s calls %6d  avg %8.1f us  total %8.1f ms" % (m.group(0) if m else n[:34], c, a, s))
PY
    break; fi
done 2>&1 | tee $O/traced.txt
rm -rf $O/trace; rm -f $T
