#!/bin/bash
# On the GPU box: full GPU suite, smoke(), the default bench line.
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
( time python -m pytest tests -m gpu -x -q -p no:cacheprovider ) > gpurun_out/$1_tests.log 2>&1; tail -6 gpurun_out/$1_tests.log
( time python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) > gpurun_out/$1_smoke.log 2>&1; tail -4 gpurun_out/$1_smoke.log
( time python bench.py ) > gpurun_out/$1_default.json 2> gpurun_out/$1_default.err; tail -3 gpurun_out/$1_default.err; wc -c gpurun_out/$1_default.json
