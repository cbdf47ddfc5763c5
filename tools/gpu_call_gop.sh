cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( time python -m pytest tests/test_gpu_gop.py -m gpu -q -p no:cacheprovider ) > gpurun_out/r03m_gop.log 2>&1; tail -25 gpurun_out/r03m_gop.log
