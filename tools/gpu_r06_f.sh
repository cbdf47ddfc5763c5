#!/bin/bash
# On the GPU box (round 6, call f): copy / kernel timeline of the host-fed frame queue; k_dec_index after the long-entry change.
cd ${GRAFT_REPO_ROOT:-/root/repo}; R=$PWD; mkdir -p gpurun_out/r06f; O=$R/gpurun_out/r06f
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-c-abi --no-other-workloads --depth 1 > $O/d1.json 2>/dev/null; python -c "import json;d=json.loads(open('$O/d1.json').read().strip().splitlines()[-1]);k=d['config']['kernel_ms_per_step'];print('depth 1 fps', d['value'], {n:k[n] for n in k if n.startswith('k_dec')})"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace -d $O/trace -o hf -- python $R/tools/host_fed_probe.py 128 4 16 > $O/probe.txt 2>&1; tail -2 $O/probe.txt | cut -c1-400
D=$(find $O/trace -name '*.db' | head -1); echo "db: $D"
python $R/tools/copy_timeline.py $D 2>&1 | tee $O/timeline.txt
