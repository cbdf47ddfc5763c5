cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time python -m pytest tests -m gpu -q --durations=15 -p no:cacheprovider ) > gpurun_out/r03a_tests.log 2>&1
tail -5 gpurun_out/r03a_tests.log
python bench.py --steps 20 --warmup 3 > gpurun_out/r03a_bench.json 2> gpurun_out/r03a_bench.err
tail -c 600 gpurun_out/r03a_bench.json
bash tools/profile_sq.sh r03a > gpurun_out/r03a_sq.log 2>&1
tail -3 gpurun_out/r03a_sq.log
