cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
timeout 1200 python -m pytest tests/test_tower_train_gpu.py tests/test_train_mode_gpu.py tests/test_train_loop_gpu.py tests/test_bench_gpu.py -q -m gpu --tb=short -x > $O/tests_c.log 2>&1; tail -12 $O/tests_c.log | cut -c1-250
bash tools/r04_bt.sh 2>&1 | tail -12
