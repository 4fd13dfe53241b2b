cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
timeout 1200 python -m pytest tests/test_tower_train_gpu.py tests/test_train_mode_gpu.py tests/test_train_gpu.py tests/test_train_loop_gpu.py tests/test_bf16_mode_gpu.py -q -m gpu --tb=short > $O/tests_b.log 2>&1; tail -25 $O/tests_b.log | cut -c1-250
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/step_trace -- python3 $GRAFT_REPO_ROOT/tools/bench_train.py --config c3 --prec bf16 --steps 12 --warmup 6 --modes graph > $GRAFT_REPO_ROOT/$O/step_trace.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/step_timeline.py $O/step_trace 25 > $O/step_timeline_bf16_graph.txt 2>&1; head -3 $O/step_timeline_bf16_graph.txt
