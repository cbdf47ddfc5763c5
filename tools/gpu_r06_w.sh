#!/bin/bash
# On the GPU box (round 6, call w): the host-fed frame queue against the number of hardware queues the runtime grants (GPU_MAX_HW_QUEUES) and the passes in flight.
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out/r06w
for q in "" 8 16 24; do for cfg in "128 6" "128 8" "64 12"; do
  set -- $cfg
  r=$( ( [ -n "$q" ] && export GPU_MAX_HW_QUEUES=$q; timeout 300 python tools/host_fed_probe.py $1 $2 36 2>/dev/null | tail -1 ) )
  echo "queues=${q:-default} batch=$1 depth=$2 $(echo "$r" | python -c "import sys,ast; d=ast.literal_eval(sys.stdin.read()); print('fps', d.get('fps'), 'GB/s', d.get('pcie_gbs_both_directions'), 'ms/pass', d.get('ms_per_pass'))" 2>/dev/null || echo "$r" | cut -c1-200)"
done; done | tee gpurun_out/r06w/host_fed_queues.txt
