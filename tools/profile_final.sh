#!/bin/bash
# On the GPU box, end of a round: rocprofv3 kernel trace (--kernel-trace --stats only) of the bench for the workloads whose kernels changed,
# summarised per kernel, and plain bench lines (no CPU baseline, no C-ABI leg) of the others.  Output: gpurun_out/prof_final/.
#   tools/profile_final.sh "<traced workloads>" "<plain workloads>"
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_final
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for wl in $1; do
  rocprofv3 --kernel-trace --stats -d $OUT/trace_$wl -o bench -- python $ROOT/bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline --no-c-abi > $OUT/bench_traced_$wl.json 2> $OUT/trace_$wl.err
  T=$(find $OUT/trace_$wl -name '*.db' | head -1)
  { echo "# rocprofv3 --kernel-trace --stats -- python bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline --no-c-abi"; echo "# bench line under the trace:"; cat $OUT/bench_traced_$wl.json; python $ROOT/tools/rocprof_summary.py $T; } > $OUT/summary_$wl.txt 2> $OUT/summary_$wl.err
  rm -rf $OUT/trace_$wl
done
for wl in $2; do
  python $ROOT/bench.py --workload $wl --steps 10 --warmup 3 --no-cpu-baseline --no-c-abi > $OUT/bench_$wl.json 2> $OUT/bench_$wl.err
done
ls -la $OUT
