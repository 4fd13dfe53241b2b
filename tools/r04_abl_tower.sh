# ablation of the fused training-tower kernels (diagnostic builds; the shipped library is rebuilt at the end)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04
mkdir -p $O
for abl in 0 1 2 4 8 15; do
  touch $R/dl-dkd_amd/csrc/tower_train.hip
  make -C $R/dl-dkd_amd/csrc TT_ABL=$abl > /dev/null 2>&1
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/abl_$abl -- python3 $R/tools/bench_train.py --config c3 --prec bf16 --steps 10 --warmup 2 --modes eager > $O/abl_$abl.log 2>&1
  cd $R
  echo "== TT_ABL=$abl"; python3 tools/kstats.py gpurun_out/r04/abl_$abl 40 | grep "tt::" | cut -c1-140
done
touch $R/dl-dkd_amd/csrc/tower_train.hip; make -C $R/dl-dkd_amd/csrc > /dev/null 2>&1
