cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
timeout 300 python -m pytest tests/test_tower_train_gpu.py -q -m gpu --tb=short 2>&1 | tail -4 | cut -c1-300
timeout 900 python tools/rk_gate_tvr.py --seeds 1 --nv 2048 --nq 4096 --steps 1500 --sigma 6.0 --out $O/rk_gate_trial.json 2>&1 | tail -24 | cut -c1-330
bash tools/r04_bt.sh 2>&1 | tail -11
