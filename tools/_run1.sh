cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 3000 python -m pytest tests -x -q -m gpu > gpurun_out/gpu_suite.log 2>&1
grep -E "passed|failed|error" gpurun_out/gpu_suite.log | tail -5
