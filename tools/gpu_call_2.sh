#!/bin/bash
# On the GPU box: full GPU suite, the default bench line (all legs), the dual-batch experiment.
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
( time python -m pytest tests -m gpu -x -q -p no:cacheprovider ) > gpurun_out/$1_tests.log 2>&1; tail -6 gpurun_out/$1_tests.log
( time python bench.py ) > gpurun_out/$1_default.json 2> gpurun_out/$1_default.err; tail -3 gpurun_out/$1_default.err; wc -c gpurun_out/$1_default.json
( time timeout 300 python tools/dual_batch.py 512 20 ) > gpurun_out/$1_dual.log 2>&1; tail -8 gpurun_out/$1_dual.log
