#!/bin/bash
# On the GPU box: the round's profile evidence -- kernel trace + PMC traffic for 1080p (512 frames), byr4-2160p (96), 1080i (512); SQ counters for 1080p.
cd ${GRAFT_REPO_ROOT:-/root/repo}
bash tools/profile_round.sh r03_1080p 512 > gpurun_out/r03_prof_1080p.log 2>&1
bash tools/profile_round.sh r03_byr4 96 --workload byr4-2160p > gpurun_out/r03_prof_byr4.log 2>&1
bash tools/profile_round.sh r03_1080i 512 --workload 1080i > gpurun_out/r03_prof_1080i.log 2>&1
bash tools/profile_sq.sh r03_after > gpurun_out/r03_sq_after.log 2>&1
ls gpurun_out/prof_r03_1080p gpurun_out/prof_r03_byr4 gpurun_out/prof_r03_1080i gpurun_out/sq_r03_after
tail -3 gpurun_out/prof_r03_1080p/summary.err
