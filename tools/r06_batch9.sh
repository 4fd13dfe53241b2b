#!/bin/bash
mkdir -p gpurun_out/r06
R=$PWD; O=$R/gpurun_out/r06
T() { name=$1; shift; timeout 1200 "$@" > $O/$name.log 2>&1; echo "$name rc=$?"; tail -4 $O/$name.log; }
T lndual python -m pytest tests/test_ln_dual_gpu.py -x -q -m gpu
T trainmode python -m pytest tests/test_train_mode_gpu.py -q -m gpu -s -k "mixed"
T bt_mixed_dual python tools/bench_train.py --config c3 --prec mixed --modes graph
DLDKD_LN_DUAL=0 T bt_mixed_nodual python tools/bench_train.py --config c3 --prec mixed --modes graph
T bt_mixed_dual2 python tools/bench_train.py --config c3 --prec mixed --modes graph
T bt_mixed_c5 python tools/bench_train.py --config c5 --prec mixed --modes graph
