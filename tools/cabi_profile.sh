#!/bin/bash
# On the GPU box: kernel trace of the C-ABI benchmark (how busy is the GPU while host threads feed it frame by frame?)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
python - <<'PY'
import sys
sys.path.insert(0, "tests")
import cfhd_testlib as T
frames, pitch = T.qbist_frames(10, 8, 1920, 1080)
open("/tmp/frames.yuy2", "wb").write(b"".join(fr.tobytes() for fr in frames))
PY
OUT=$ROOT/gpurun_out/prof_cabi; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o cabi -- $ROOT/tools/_build/cabi_bench 1920 1080 /tmp/frames.yuy2 8 1.5 ${REG:-0} ${DEC:-8} ${WORK:-8} > $OUT/line.json 2> $OUT/err.txt
T=$(find $OUT/trace -name '*.db' | head -1)
python $ROOT/tools/rocprof_summary.py $T > $OUT/summary.txt 2>> $OUT/err.txt
cat $OUT/line.json; head -40 $OUT/summary.txt
rm -rf $OUT/trace
