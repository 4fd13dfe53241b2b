#!/bin/bash
cd /root/repo
timeout 900 python -m pytest tests/test_tower_train_gpu.py tests/test_train_mode_gpu.py -q -m gpu -x > gpurun_out/r04_tt_tests.log 2>&1; grep "passed\|failed" gpurun_out/r04_tt_tests.log | tail -1
R=$PWD; O=$R/gpurun_out/r04tt; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/train -- python3 $R/tools/bench_train.py --config c3 --prec bf16 --steps 20 --warmup 3 --modes eager > $O/train.log 2>&1
python3 $R/tools/kstats.py $O/train 16 | grep "tt::\|dw_\|layernorm"
cd $R
for c in c3 c5; do python tools/bench_train.py --config $c --prec bf16 --steps 40 --warmup 10 --modes graph 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config'], {k:round(v['stream_ms_median'],3) for k,v in d.items() if isinstance(v,dict)})"; done
find $O -name "*kernel_trace.csv" -delete
