#!/bin/bash
# On the GPU box (round 5, call i): SQ counters of the single-pass arrangement (k_dec_index_emit, k_dec_scatter) and, in the same passes, of everything else:
# what do the waves of the tile pass wait for?
cd ${GRAFT_REPO_ROOT:-/root/repo}; R=$PWD
CFHD_AMD_DEC=emit bash tools/profile_sq.sh r05i_emit --depth 1 > /dev/null 2>&1
bash tools/profile_sq.sh r05i_tiles --depth 1 > /dev/null 2>&1
for t in r05i_emit r05i_tiles; do
  for p in 0 1 2; do grep "k_dec_scatter\|k_dec_tiles\|k_dec_index" gpurun_out/sq_$t/summary_p$p.txt | grep "SQ_" | sed 's/_ZN4cfhd3dev[0-9]*\(k_dec_[a-z_]*\)E[A-Za-z0-9_]*/\1/' | awk -v T=$t '{printf "%s %-18s %-24s %s\n", T, $1, $2, $5}'; done
done
