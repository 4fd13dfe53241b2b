#!/bin/bash
# same-box A/B: number of HSA hardware queues the HIP runtime multiplexes its streams onto (GPU_MAX_HW_QUEUES, default 4)
cd /root/repo
for q in 4 8 4 8 6; do
for c in c3 c5; do
GPU_MAX_HW_QUEUES=$q timeout 300 python tools/bench_train.py --config $c --prec bf16 --steps 40 --warmup 10 --modes graph 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('hwq=$q', d['config'], {k:round(v['stream_ms_median'],3) for k,v in d.items() if isinstance(v,dict)})"
done; done
