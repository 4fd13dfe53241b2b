#!/bin/bash
# round 6, third GPU batch: dual LayerNorm on the video stream, mixed mode with the fused bf16 backward, eval_epoch wall at C2
mkdir -p gpurun_out/r06
R=$PWD; O=$R/gpurun_out/r06
T() { name=$1; shift; timeout 900 "$@" > $O/$name.log 2>&1; echo "$name rc=$?"; tail -4 $O/$name.log; }
T lndual python -m pytest tests/test_ln_dual_gpu.py -x -q -m gpu
T trainmode python -m pytest tests/test_train_mode_gpu.py -q -m gpu -s
T bf16mode python -m pytest tests/test_bf16_mode_gpu.py tests/test_train_loop_gpu.py tests/test_train_gpu.py tests/test_tower_train_gpu.py -x -q -m gpu
T bt_mixed python tools/bench_train.py --config c3 --prec mixed --modes graph
T bt_mixed_c5 python tools/bench_train.py --config c5 --prec mixed --modes graph
T bt_bf16_dual python tools/bench_train.py --config c3 --prec bf16 --modes graph
DLDKD_LN_DUAL=0 T bt_bf16_nodual python tools/bench_train.py --config c3 --prec bf16 --modes graph
T bt_bf16_dual2 python tools/bench_train.py --config c3 --prec bf16 --modes graph
T evalc2 python tools/bench_eval_epoch_c2.py --profile $O/eval_epoch_c2_cached_cprofile.txt
