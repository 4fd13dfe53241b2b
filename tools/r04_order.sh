#!/bin/bash
cd /root/repo
R=$PWD; O=$R/gpurun_out/r04o; rm -rf $O; mkdir -p $O
for rev in 0 1 0 1; do
for c in c3 c5; do DLDKD_BWD_REV=$rev timeout 300 python tools/bench_train.py --config $c --prec bf16 --steps 40 --warmup 10 --modes graph 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('rev=$rev', d['config'], {k:round(v['stream_ms_median'],3) for k,v in d.items() if isinstance(v,dict)})"; done
done
cd /tmp && export TMPDIR=/tmp
export DLDKD_BWD_REV=1
timeout 300 rocprofv3 --kernel-trace --hip-trace --output-format csv -d $O/rev_c5 -- python3 $R/tools/bench_train.py --config c5 --prec bf16 --steps 12 --warmup 3 --modes graph > $O/rev_c5.log 2>&1
find $O -name "*agent_info.csv" -delete
ls -la $O/rev_c5/*/ | head
