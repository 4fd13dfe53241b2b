cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu --tb=short > $O/tests_full.log 2>&1; tail -15 $O/tests_full.log | cut -c1-250
bash tools/r04_bt.sh 2>&1 | tail -11
