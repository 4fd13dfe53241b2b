#!/bin/bash
# round 6, fourth GPU batch: two-plane forward of the mixed mode, skip-zero-rows A/B of the resident encode, eval wall
mkdir -p gpurun_out/r06
R=$PWD; O=$R/gpurun_out/r06
T() { name=$1; shift; timeout 900 "$@" > $O/$name.log 2>&1; echo "$name rc=$?"; tail -4 $O/$name.log; }
T x2 python -m pytest tests/test_gemm_x3_gpu.py -x -q -m gpu
T trainmode python -m pytest tests/test_train_mode_gpu.py -q -m gpu -s -k "mixed"
T towerseq python -m pytest tests/test_tower_seq_gpu.py tests/test_eval_gpu.py tests/test_rk_gate_gpu.py -x -q -m gpu -k "not three_seeds"
T bt_mixed_x2 python tools/bench_train.py --config c3 --prec mixed --modes graph
DLDKD_MIXED_FORWARD=fp32 T bt_mixed_x3 python tools/bench_train.py --config c3 --prec mixed --modes graph
T bt_mixed_c5 python tools/bench_train.py --config c5 --prec mixed --modes graph
for v in 1 0 1 0; do DLDKD_SKIP_ZERO_ROWS=$v ENC_BATCH=1024 T enc_skip$v python tools/prof_encode.py resident; done
T evalc2 python tools/bench_eval_epoch_c2.py --no-oracle
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/graph_c3_mixed2 -- python3 $R/tools/bench_train.py --config c3 --prec mixed --steps 12 --warmup 3 --modes graph > $O/graph_c3_mixed2.log 2>&1
python3 $R/tools/step_timeline.py $O/graph_c3_mixed2 20 > $O/step_timeline_c3_mixed_graph.txt 2>&1
head -1 $O/step_timeline_c3_mixed_graph.txt
cd $R
find $O -name "*kernel_trace.csv" -size +3M -delete
find $O -name "*agent_info.csv" -delete
