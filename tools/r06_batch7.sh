#!/bin/bash
mkdir -p gpurun_out/r06
R=$PWD; O=$R/gpurun_out/r06
T() { name=$1; shift; timeout 1200 "$@" > $O/$name.log 2>&1; echo "$name rc=$?"; tail -4 $O/$name.log; }
T planes python -m pytest tests/test_gemm_x3_gpu.py -x -q -m gpu
T trainmode python -m pytest tests/test_train_mode_gpu.py -q -m gpu -s -k "mixed"
T bt_mixed_planes python tools/bench_train.py --config c3 --prec mixed --modes graph
DLDKD_MIXED_PLANES=0 T bt_mixed_noplanes python tools/bench_train.py --config c3 --prec mixed --modes graph
T bt_mixed_c5 python tools/bench_train.py --config c5 --prec mixed --modes graph
T gt_mixed python tools/graph_timeline.py c3 sync mixed
T trainsuite python -m pytest tests/test_train_gpu.py tests/test_train_loop_gpu.py tests/test_tower_train_gpu.py tests/test_bf16_mode_gpu.py -x -q -m gpu
