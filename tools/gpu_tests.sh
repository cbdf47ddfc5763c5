#!/bin/bash
# On the GPU box: a selection of the GPU tests.  tools/gpu_tests.sh <tag> "<pytest -k expression>"
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
( time python -m pytest tests -m gpu -q -p no:cacheprovider -k "$2" ) > gpurun_out/$1_tests.log 2>&1; tail -25 gpurun_out/$1_tests.log
