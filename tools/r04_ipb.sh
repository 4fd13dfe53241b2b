#!/bin/bash
cd /root/repo
python -m pytest tests/test_bf16_mode_gpu.py -q -s -m gpu -k "two_accumulator or skips_the_padding or bf16_rows" > gpurun_out/r04_ipb.log 2>&1; grep -v "^$" gpurun_out/r04_ipb.log | grep "rel l2\|passed\|failed\|Error" | head -20


