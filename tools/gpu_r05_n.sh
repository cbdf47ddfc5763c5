#!/bin/bash
# On the GPU box (round 5, call n): the frame queue in its forms -- a pass queued as a whole on the batch's streams (with the sample bytes sent ahead, sized by the
# previous pass) | the blocking pass on a host thread per batch (round 4, CFHD_AMD_QUEUE=thread) -- at one to four steps in flight.
cd ${GRAFT_REPO_ROOT:-/root/repo}; R=$PWD; mkdir -p gpurun_out/r05n; O=gpurun_out/r05n
B="--steps 30 --warmup 4 --no-cpu-baseline --no-c-abi --no-other-workloads"
for q in events unordered thread; do for d in 1 2 3 4; do
  CFHD_AMD_QUEUE=$q python bench.py $B --depth $d > $O/${q}_$d.json 2> $O/${q}_$d.err
  python -c "import json;d=json.loads(open('$O/${q}_$d.json').read().strip().splitlines()[-1]);print('queue $q depth $d', d['value'], 'fps', d['ms_per_step'], 'ms per step', d['config']['parity']['samples_equal_reference_encoder'], d['config']['parity']['decoded_frames_in_dither_interval'])"
done; done
