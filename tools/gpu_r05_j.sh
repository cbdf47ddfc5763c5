#!/bin/bash
# On the GPU box (round 5, call j): where the wall time of a synchronous CFHD_DecodeSample / CFHD_EncodeSample goes (CFHD_AMD_PROFILE=1: host stages per call) and what the GPU
# does during it (kernel trace of the same program: launches per call, kernel time per call, the gaps between them).
cd ${GRAFT_REPO_ROOT:-/root/repo}; R=$PWD; mkdir -p gpurun_out/r05j; O=gpurun_out/r05j
python - <<'PY'
import sys
sys.path.insert(0, "tests")
import cfhd_testlib as T
frames, pitch = T.qbist_frames(10, 8, 1920, 1080, T.PIX_YUY2)
with open("/tmp/frames.yuy2", "wb") as f:
    for fr in frames: f.write(fr.reshape(1080, pitch)[:, :3840].tobytes())
PY
echo "== stages, plain buffers"; CFHD_AMD_PROFILE=1 CFHD_AMD_DEVICE=0 tools/_build/cabi_bench 1920 1080 /tmp/frames.yuy2 8 1.0 0 1 2 2>&1 | grep -a "cfhd_amd\]\|sync" | head -8
echo "== stages, registered buffers"; CFHD_AMD_PROFILE=1 CFHD_AMD_DEVICE=0 tools/_build/cabi_bench 1920 1080 /tmp/frames.yuy2 8 1.0 1 1 2 2>&1 | grep -a "cfhd_amd\]\|sync" | head -8
cd /tmp; export TMPDIR=/tmp
CFHD_AMD_DEVICE=0 timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --stats -d $R/$O/trace -o sync -- $R/tools/_build/cabi_bench 1920 1080 /tmp/frames.yuy2 8 0.25 0 1 2 > $R/$O/trace.log 2>&1
cd $R
python tools/rocprof_summary.py $O/trace 2>/dev/null | head -60 > $O/trace_summary.txt
python - <<'PY'
import sqlite3, glob, sys
db = sorted(glob.glob("gpurun_out/r05j/trace/**/*.db", recursive=True))
if not db: print("no db"); sys.exit(0)
c = sqlite3.connect(db[0])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = c.execute(f"select k.start, k.end, s.kernel_name from {kd} k join {sym} s on k.kernel_id = s.id order by k.start").fetchall()
print("dispatches", len(rows))
# the decode phase: sequences starting with k_dec_parse
import collections
seqs = []; cur = None
for st, en, nm in rows:
    short = nm.split("(")[0]
    if "k_dec_parse" in short:
        if cur: seqs.append(cur)
        cur = []
    if cur is not None: cur.append((st, en, short))
if cur: seqs.append(cur)
seqs = [s for s in seqs if len(s) > 5]
print("decode sequences", len(seqs))
if seqs:
    mid = seqs[len(seqs)//2 - 20: len(seqs)//2 + 20]
    import statistics as S
    print("launches per decode", S.median(len(s) for s in mid))
    print("kernel time per decode us", S.median(sum(e - s for s, e, _ in q) for q in mid) / 1e3)
    print("first start to last end us", S.median(q[-1][1] - q[0][0] for q in mid) / 1e3)
    print("parse to next parse us", S.median(b[0][0] - a[0][0] for a, b in zip(mid, mid[1:])) / 1e3)
    q = mid[0]; t0 = q[0][0]
    for st, en, nm in q: print(f"  {nm[-40:]:40s} start {(st - t0)/1e3:8.1f} us  dur {(en - st)/1e3:7.1f} us")
PY
