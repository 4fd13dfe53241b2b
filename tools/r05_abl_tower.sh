# upper bound of sharing the training towers' weight fragments through LDS: the row kernels with their weight loads taken out
# (TT_ABL=2: the 24-slot register ring is filled once) against the shipped build, same box; the shipped library is rebuilt at the end
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05abl
mkdir -p $O
for abl in 0 2 0 2; do
  touch $R/dl-dkd_amd/csrc/tower_train.hip
  make -C $R/dl-dkd_amd/csrc TT_ABL=$abl > /dev/null 2>&1
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/abl_$abl -- python3 $R/tools/bench_train.py --config c3 --prec bf16 --steps 10 --warmup 2 --modes eager > $O/abl_$abl.log 2>&1
  cd $R
  echo "== TT_ABL=$abl"; python3 tools/kstats.py gpurun_out/r05abl/abl_$abl 40 | grep "tt::" | cut -c1-140
  python tools/bench_train.py --config c3 --prec bf16 --steps 30 --warmup 8 --modes graph 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config'], {k:round(v['stream_ms_median'],3) for k,v in d.items() if isinstance(v,dict)})"
  rm -rf $O/abl_$abl
done
touch $R/dl-dkd_amd/csrc/tower_train.hip; make -C $R/dl-dkd_amd/csrc > /dev/null 2>&1
