#!/bin/bash
mkdir -p gpurun_out/r06
R=$PWD; O=$R/gpurun_out/r06
T() { name=$1; shift; timeout 1200 "$@" > $O/$name.log 2>&1; echo "$name rc=$?"; tail -4 $O/$name.log; }
T trainmode python -m pytest tests/test_train_mode_gpu.py -q -m gpu -s -k "mixed"
T bt_mixed python tools/bench_train.py --config c3 --prec mixed --modes graph
T bt_mixed_c5 python tools/bench_train.py --config c5 --prec mixed --modes graph
T gt_bf16 python tools/graph_timeline.py c3 sync bf16
T gt_mixed python tools/graph_timeline.py c3 sync mixed
T evalc2 python tools/bench_eval_epoch_c2.py --no-oracle --profile $O/eval_epoch_c2_cached_cprofile.txt
T regress python -m pytest tests/test_eval_gpu.py tests/test_train_gpu.py tests/test_train_loop_gpu.py tests/test_api_edges_gpu.py tests/test_overflow_guard_gpu.py tests/test_shard_gpu.py tests/test_simpool_gpu.py tests/test_encoder_gpu.py -x -q -m gpu
