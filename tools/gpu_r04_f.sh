#!/bin/bash
# On the GPU box (round 4, call f): the group tests (GPU entropy stage for groups, rate feedback), the strip tests with the wide plane segments, then the evidence for
# the 2160p line (plain / traced / PMC) and the kernel trace of the default 1080p line as it runs (three steps in flight).
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
( time timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "gop or group or strip_kernels or rate_feedback or bench_sizes" ) > gpurun_out/$1_tests.log 2>&1; tail -4 gpurun_out/$1_tests.log
bash tools/profile_round.sh $1_2160p 128 --workload 2160p --depth 1 > gpurun_out/$1_profile_2160p.log 2>&1; head -c 1500 gpurun_out/prof_$1_2160p/bench_plain.json | tr ',' '\n' | grep -i "value\|ms_per_step\|k_fwd_plane\|k_inv_plane\|frac" | head -20
bash tools/profile_round.sh $1_1080p_depth3 512 --depth 3 > gpurun_out/$1_profile_1080p_depth3.log 2>&1; head -c 600 gpurun_out/prof_$1_1080p_depth3/bench_plain.json
