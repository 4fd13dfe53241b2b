#!/bin/bash
cd /root/repo
timeout 900 python -m pytest tests/test_tower_train_gpu.py tests/test_train_mode_gpu.py tests/test_train_loop_gpu.py tests/test_train_gpu.py tests/test_bf16_mode_gpu.py -q -m gpu -x > gpurun_out/r04_count_tests.log 2>&1; grep "passed\|failed" gpurun_out/r04_count_tests.log | tail -2
R=$PWD; O=$R/gpurun_out/r04c; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for cfg in c3 c5; do
rocprofv3 --kernel-trace --output-format csv -d $O/graph_$cfg -- python3 $R/tools/bench_train.py --config $cfg --prec bf16 --steps 12 --warmup 3 --modes graph > $O/graph_$cfg.log 2>&1
python3 $R/tools/step_timeline.py $O/graph_$cfg 20 > $O/step_timeline_${cfg}.txt 2>&1; head -1 $O/step_timeline_${cfg}.txt
done
cd $R
for c in c3 c5; do python tools/bench_train.py --config $c --prec bf16 --steps 40 --warmup 10 --modes graph 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config'], {k:round(v['stream_ms_median'],3) for k,v in d.items() if isinstance(v,dict)})"; done
find $O -name "*agent_info.csv" -delete
