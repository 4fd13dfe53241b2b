# where the training row kernels' time goes: diagnostic builds with parts taken out (TT_ABL bits: 1 no Philox, 2 no weight loads,
# 4 no LayerNorm-gradient column sums, 8 no MFMAs), eager kernel averages under rocprofv3; the shipped library is rebuilt at the end
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05abl2
mkdir -p $O
for abl in 0 1 4 8 10 15; do
  touch $R/dl-dkd_amd/csrc/tower_train.hip
  make -C $R/dl-dkd_amd/csrc TT_ABL=$abl > /dev/null 2>&1
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/abl_$abl -- python3 $R/tools/bench_train.py --config c3 --prec bf16 --steps 8 --warmup 2 --modes eager > $O/abl_$abl.log 2>&1
  cd $R
  echo "== TT_ABL=$abl"; python3 tools/kstats.py gpurun_out/r05abl2/abl_$abl 40 | grep "tt::" | cut -c1-140
  rm -rf $O/abl_$abl
done
touch $R/dl-dkd_amd/csrc/tower_train.hip; make -C $R/dl-dkd_amd/csrc > /dev/null 2>&1
