#!/bin/bash
# round 6, first GPU batch: new tests + mixed-mode timing
mkdir -p gpurun_out/r06
T() { name=$1; shift; timeout 900 "$@" > gpurun_out/r06/$name.log 2>&1; echo "$name rc=$?"; tail -4 gpurun_out/r06/$name.log; }
T comm python -m pytest tests/test_comm_gpu.py -x -q -m gpu -k "host_wait"
T launch python -m pytest tests/test_bench_gpu.py -x -q -m gpu -k "more_gpus or self_launched"
T overflow python -m pytest tests/test_overflow_guard_gpu.py -x -q -m gpu -s
T shard python -m pytest tests/test_shard_gpu.py -x -q -m gpu
T tower python -m pytest tests/test_tower_seq_gpu.py tests/test_eval_gpu.py -x -q -m gpu
T mixed python -m pytest tests/test_train_mode_gpu.py -x -q -m gpu -s -k "mixed"
T bt_mixed python tools/bench_train.py --config c3 --prec mixed --modes graph
T bt_mixed_c5 python tools/bench_train.py --config c5 --prec mixed --modes graph
T bt_fp32 python tools/bench_train.py --config c3 --prec fp32 --modes graph
T bt_bf16 python tools/bench_train.py --config c3 --prec bf16 --modes graph
