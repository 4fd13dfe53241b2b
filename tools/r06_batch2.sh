#!/bin/bash
# round 6, second GPU batch: dual LayerNorm + mixed-mode profile
mkdir -p gpurun_out/r06
R=$PWD; O=$R/gpurun_out/r06
T() { name=$1; shift; timeout 900 "$@" > $O/$name.log 2>&1; echo "$name rc=$?"; tail -4 $O/$name.log; }
T lndual python -m pytest tests/test_ln_dual_gpu.py -x -q -m gpu
T bf16mode python -m pytest tests/test_bf16_mode_gpu.py tests/test_train_mode_gpu.py tests/test_train_loop_gpu.py -x -q -m gpu
T bt_bf16_dual python tools/bench_train.py --config c3 --prec bf16 --modes graph
DLDKD_LN_DUAL=0 T bt_bf16_nodual python tools/bench_train.py --config c3 --prec bf16 --modes graph
T bt_c5_dual python tools/bench_train.py --config c5 --prec bf16 --modes graph
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/train_c3_mixed -- python3 $R/tools/bench_train.py --config c3 --prec mixed --steps 20 --warmup 3 --modes eager > $O/train_c3_mixed.log 2>&1
f=$(ls -t $(grep -l "dldkd::" $(find $O/train_c3_mixed -name "*kernel_stats.csv")) | head -1); [ -n "$f" ] && cp $f $O/train_c3_mixed_kernel_stats.csv
for pr in mixed bf16; do
rocprofv3 --kernel-trace --output-format csv -d $O/graph_c3_$pr -- python3 $R/tools/bench_train.py --config c3 --prec $pr --steps 12 --warmup 3 --modes graph > $O/graph_c3_$pr.log 2>&1
python3 $R/tools/step_timeline.py $O/graph_c3_$pr 20 > $O/step_timeline_c3_${pr}_graph.txt 2>&1
head -1 $O/step_timeline_c3_${pr}_graph.txt
done
cd $R
find $O -name "*kernel_trace.csv" -size +3M -delete
find $O -name "*agent_info.csv" -delete
