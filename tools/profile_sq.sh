#!/bin/bash
# On the GPU box: SQ counter passes of a bench configuration (separate --pmc runs, never combined with a trace): instruction mix,
# wave cycles, where waves wait.  Tells an issue-bound kernel (ACTIVE_INST_* ~ WAVE_CYCLES / waves per SIMD) from a latency-bound one (WAIT_ANY).
#   tools/profile_sq.sh <tag> [bench.py arguments]
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/sq_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters_available.txt 2>&1
PASSES=(
 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM"
 "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM"
 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_FLAT SQ_WAVES SQ_WAVE_CYCLES"
)
i=0
for P in "${PASSES[@]}"; do
  rocprofv3 --pmc $P -d $OUT/p$i -o bench -- python $ROOT/bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline --no-c-abi --no-other-workloads > $OUT/bench_p$i.json 2> $OUT/p$i.err
  D=$(find $OUT/p$i -name '*.db' | head -1)
  [ -n "$D" ] && python $ROOT/tools/rocprof_summary.py $D > $OUT/summary_p$i.txt 2>> $OUT/p$i.err
  rm -rf $OUT/p$i
  i=$((i+1))
done
ls -la $OUT
