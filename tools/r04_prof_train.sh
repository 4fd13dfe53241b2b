set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/train_bf16_fused -- python3 $R/tools/bench_train.py --config c3 --prec bf16 --steps 20 --warmup 3 --modes eager > $O/train_bf16_fused.log 2>&1
cd $R
python3 tools/kstats.py gpurun_out/r04/train_bf16_fused 40
