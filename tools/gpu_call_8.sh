#!/bin/bash
# On the GPU box: decoder-side GPU tests, then the 1080p bench with the tile pass split (default) and unsplit, 1080i, 2160p.
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
( time python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "batched or bench_size or interlaced or concurrent or fuzz or half or decode_service or b64a_decode or rg48" ) > gpurun_out/$1_tests.log 2>&1; tail -5 gpurun_out/$1_tests.log
bash tools/gpu_probe.sh $1 "CFHD_X=0" "CFHD_AMD_TILES_SPLIT=0" | cut -c1-520
for WL in 1080i 2160p; do
python bench.py --workload $WL --steps 10 --warmup 3 --no-cpu-baseline --no-c-abi 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$WL', d['value'], d['config']['parity'])"
done
