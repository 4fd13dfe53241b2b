cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
timeout 900 python tools/rk_gate_tvr.py --seeds 1 --nv 1024 --nq 2048 --steps 300 --sigma 6.0 --out $O/rk_gate_trial.json 2>&1 | tail -14 | cut -c1-400
timeout 600 python tools/ablation_simpool_ragged.py --iters 12 > $O/ablation_simpool_ragged.json 2>$O/abl_rag.err; tail -30 $O/ablation_simpool_ragged.json
