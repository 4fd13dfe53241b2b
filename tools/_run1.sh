cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 200 python tools/x3_check.py 3072 131072 2>&1 | tail -4
timeout 3000 python -m pytest tests -x -q -m gpu > gpurun_out/gpu_suite.log 2>&1
grep -E "passed|failed|error|Error" gpurun_out/gpu_suite.log | tail -8
