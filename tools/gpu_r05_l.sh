#!/bin/bash
# On the GPU box (round 5, call l): the whole GPU suite on the final tree, smoke(), then (call m) the default bench line, the profile of one step at a time, depth / batch side lines.
cd ${GRAFT_REPO_ROOT:-/root/repo}; R=$PWD; mkdir -p gpurun_out/r05l; O=gpurun_out/r05l
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gpu_suite.log 2>&1; echo "suite rc $?"; tail -6 $O/gpu_suite.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc $?"; tail -2 $O/smoke.log
