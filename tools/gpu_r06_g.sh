#!/bin/bash
# On the GPU box (round 6, call g): the host-fed frame queue with the device's copy lanes (one copy per direction at a time) against copies on the passes' own streams.
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out/r06g; O=gpurun_out/r06g
python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "fed_from_host or frame_queue" 2>&1 | tail -2
for lanes in 1 0; do for cfg in "128 4 24" "256 3 18" "64 6 36" "128 6 36" "512 3 12"; do
  echo "lanes=$lanes $cfg: $(CFHD_AMD_COPY_LANES=$lanes python tools/host_fed_probe.py $cfg 2>/dev/null | grep -o "'fps': [0-9.]*, 'frames_per_pass': [0-9]*, 'passes_in_flight': [0-9]*, 'passes_timed': [0-9]*, 'ms_per_pass': [0-9.]*")"
done; done 2>&1 | tee $O/host_fed_lanes.txt
for q in 16; do echo "GPU_MAX_HW_QUEUES=$q lanes=1 128 4 24: $(GPU_MAX_HW_QUEUES=$q python tools/host_fed_probe.py 128 4 24 2>/dev/null | grep -o "'fps': [0-9.]*")"; done | tee -a $O/host_fed_lanes.txt
