#!/bin/bash
# On the GPU box (round 4, call g): the Bayer tests (k_fwd_bayer_strip forced on small batches and by size at 3840x2160), then the evidence for the byr4-2160p line.
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
( time timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "bayer or byr4 or byr5 or other_configurations" ) > gpurun_out/$1_tests.log 2>&1; tail -4 gpurun_out/$1_tests.log
bash tools/profile_round.sh $1_byr4 96 --workload byr4-2160p --depth 1 > gpurun_out/$1_profile_byr4.log 2>&1; head -c 3000 gpurun_out/prof_$1_byr4/bench_plain.json | tr ',' '\n' | grep -i "value\|ms_per_step\|k_fwd\|k_unpack\|frac\|k_ent" | head -20
