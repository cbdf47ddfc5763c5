#!/bin/bash
# On the GPU box: a selection of GPU tests, then bench probes under env settings.  tools/gpu_call_3.sh <tag> "<-k expr>" "<env1>" ...
TAG=$1; K=$2; shift 2
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
( time python -m pytest tests -m gpu -q -p no:cacheprovider -k "$K" ) > gpurun_out/${TAG}_tests.log 2>&1; tail -8 gpurun_out/${TAG}_tests.log
bash tools/gpu_probe.sh $TAG "$@" | cut -c1-400
