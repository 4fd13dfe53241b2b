#!/bin/bash
R=$PWD; O=$R/gpurun_out/r06p; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/train_c3_mixed -- python3 $R/tools/bench_train.py --config c3 --prec mixed --steps 20 --warmup 3 --modes eager > $O/train_c3_mixed.log 2>&1
f=$(ls -t $(grep -l "dldkd::" $(find $O/train_c3_mixed -name "*kernel_stats.csv")) | head -1); [ -n "$f" ] && cp $f $O/train_c3_mixed_kernel_stats.csv
rocprofv3 --kernel-trace --output-format csv -d $O/graph_c3_mixed -- python3 $R/tools/bench_train.py --config c3 --prec mixed --steps 12 --warmup 3 --modes graph > $O/graph_c3_mixed.log 2>&1
python3 $R/tools/step_timeline.py $O/graph_c3_mixed 10 > $O/step_timeline_c3_mixed_graph.txt 2>&1
head -70 $O/step_timeline_c3_mixed_graph.txt | cut -c1-130
python3 $R/tools/kstats.py $O/train_c3_mixed 30 | cut -c1-150
cd $R
find $O -name "*kernel_trace.csv" -size +3M -delete
