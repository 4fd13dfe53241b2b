#!/bin/bash
cd /root/repo
python -m pytest tests/test_tower_seq_gpu.py tests/test_eval_gpu.py tests/test_rk_gate_gpu.py -q -m gpu > gpurun_out/r04_k5_tests.log 2>&1; tail -8 gpurun_out/r04_k5_tests.log
python tools/rk_gate_tvr.py --seeds 3 --steps 1500 --out gpurun_out/rk_gate_r04.json > gpurun_out/rk_gate_r04.log 2>&1; tail -8 gpurun_out/rk_gate_r04.log
