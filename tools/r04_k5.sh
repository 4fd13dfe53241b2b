#!/bin/bash
# K5 with bf16 h0: tests + resident encode timing, bf16 h0 vs fp32 h0 (same box)
cd /root/repo
python -m pytest tests/test_tower_seq_gpu.py tests/test_eval_gpu.py -q -m gpu -x > gpurun_out/r04_k5_tests.log 2>&1; tail -15 gpurun_out/r04_k5_tests.log
cd /tmp && export TMPDIR=/tmp
for h16 in 1 0 1 0; do
DLDKD_H0_BF16=$h16 ENC_BATCH=1024 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r04_k5_prof_$h16 -- python3 /root/repo/tools/prof_encode.py resident > /root/repo/gpurun_out/r04_k5_enc_$h16.log 2>&1
tail -2 /root/repo/gpurun_out/r04_k5_enc_$h16.log
python3 /root/repo/tools/kstats.py /root/repo/gpurun_out/r04_k5_prof_$h16 4
done
find /root/repo/gpurun_out -name "*kernel_trace.csv" -size +4M -delete
