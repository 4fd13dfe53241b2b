#!/bin/bash
cd /root/repo
python -m pytest tests/test_train_loop_gpu.py tests/test_train_mode_gpu.py -q -m gpu -x > gpurun_out/r04_step_tests.log 2>&1; tail -4 gpurun_out/r04_step_tests.log | head -3
for c in c3 c5; do python tools/bench_train.py --config $c --prec bf16 --steps 40 --warmup 10 --modes graph 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config'], {k:round(v['stream_ms_median'],3) for k,v in d.items() if isinstance(v,dict)})"; done
