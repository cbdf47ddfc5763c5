#!/bin/bash
# On the GPU box: the evidence for one bench configuration -- plain bench line, rocprofv3 kernel trace (--kernel-trace --stats), and the two
# PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, never combined with a trace) -- summarised into gpurun_out/prof_<tag>/.
#   tools/profile_round.sh <tag> <frames per launch> [bench.py arguments, e.g. --workload 2160p]
set -u
TAG=$1; FRAMES=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $ROOT/bench.py "$@" --steps 10 --warmup 2 --no-cpu-baseline --no-c-abi --no-other-workloads > $OUT/bench_plain.json 2> $OUT/bench_plain.err
rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- python $ROOT/bench.py "$@" --steps 5 --warmup 2 --no-cpu-baseline --no-c-abi --no-other-workloads > $OUT/bench_traced.json 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o bench -- python $ROOT/bench.py "$@" --steps 3 --warmup 1 --no-cpu-baseline --no-c-abi --no-other-workloads > /dev/null 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE -d $OUT/write -o bench -- python $ROOT/bench.py "$@" --steps 3 --warmup 1 --no-cpu-baseline --no-c-abi --no-other-workloads > /dev/null 2> $OUT/write.err
T=$(find $OUT/trace -name '*.db' | head -1); F=$(find $OUT/fetch -name '*.db' | head -1); W=$(find $OUT/write -name '*.db' | head -1)
{
  echo "# python bench.py $* (plain run, then under rocprofv3 --kernel-trace --stats, then --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs)"
  echo "# note: under rocprofv3 the runtime copies with blit kernels (__amd_rocclr_copyBuffer.kd in the list below) instead of the SDMA engines.  The download of the previous step's"
  echo "#       samples (4.5 ms as a blit) then shares the CUs with whatever runs beside it and can stretch that kernel in the trace (k_dec_index: its minimum is its HIP-event time of the plain run)."
  echo "#       The launch times of the bench line (config.kernel_ms_*) are HIP-event times of the plain run."
  echo "# bench line of the plain run:"; cat $OUT/bench_plain.json
  echo "# bench line under the kernel trace:"; cat $OUT/bench_traced.json
  python $ROOT/tools/rocprof_summary.py $T $F $W
} > $OUT/summary.txt 2> $OUT/summary.err
WL=1080p; for a in "$@"; do [ "${prev:-}" = "--workload" ] && WL=$a; prev=$a; done
python $ROOT/tools/rocprof_summary.py --traffic-json $OUT/pmc_traffic.json --frames $FRAMES --workload $WL $F $W 2>> $OUT/summary.err
rm -rf $OUT/trace $OUT/fetch $OUT/write       # the databases are large; the summaries are what travels back
ls -la $OUT
