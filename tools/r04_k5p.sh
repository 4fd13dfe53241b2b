#!/bin/bash
cd /root/repo
python -m pytest tests/test_tower_seq_gpu.py tests/test_eval_gpu.py tests/test_rk_gate_gpu.py -q -m gpu -x > gpurun_out/r04_k5p_tests.log 2>&1; tail -4 gpurun_out/r04_k5p_tests.log | cut -c1-200
cd /tmp && export TMPDIR=/tmp
ENC_BATCH=1024 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r04_k5p_prof -- python3 /root/repo/tools/prof_encode.py resident > /root/repo/gpurun_out/r04_k5p_enc.log 2>&1
tail -1 /root/repo/gpurun_out/r04_k5p_enc.log
python3 /root/repo/tools/kstats.py /root/repo/gpurun_out/r04_k5p_prof 3
find /root/repo/gpurun_out/r04_k5p_prof -name "*kernel_trace.csv" -size +3M -delete
