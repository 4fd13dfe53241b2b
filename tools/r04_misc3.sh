cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
timeout 1500 python tools/rk_gate_tvr.py --seeds 3 --nv 4096 --nq 8192 --steps 1500 --sigma 6.0 --out $O/rk_gate.json 2>&1 | tail -8 | cut -c1-330
# K1: what the in-loop max-pool costs (diagnostic build), same box as the packing upper bound
make -C dl-dkd_amd/csrc DIAG=1 -j8 > /dev/null 2>&1
DLDKD_SIMPOOL_ABLATE=0 timeout 300 python tools/ablation_simpool_ragged.py --iters 12 > $O/ablation_simpool_ragged_diag0.json 2>$O/abl_rag.err
DLDKD_SIMPOOL_ABLATE=1 timeout 300 python tools/ablation_simpool_ragged.py --iters 12 > $O/ablation_simpool_ragged_diag1.json 2>>$O/abl_rag.err
grep -H "ms_median\|time_saved" $O/ablation_simpool_ragged_diag*.json
touch dl-dkd_amd/csrc/simpool_eval.hip; make -C dl-dkd_amd/csrc -j8 > /dev/null 2>&1
