# waves (32-row groups) per workgroup of the four training row kernels: 4 (shipped so far), 2, 1 - fewer waves share a CU's address
# coalescer when a kernel runs alone; eager kernel averages under rocprofv3 and the replayed steps
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05wpb
mkdir -p $O
for v in 4 2 1 4 2; do
  touch $R/dl-dkd_amd/csrc/tower_train.hip
  make -C $R/dl-dkd_amd/csrc TT_WPB=$v > /dev/null 2>&1
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/p -- python3 $R/tools/bench_train.py --config c3 --prec bf16 --steps 10 --warmup 2 --modes eager > $O/p.log 2>&1
  cd $R
  echo "== TT_WPB=$v"; python3 tools/kstats.py gpurun_out/r05wpb/p 70 | grep -E "tt::[fb]" | cut -c1-150
  for c in c3 c5; do python tools/bench_train.py --config $c --prec bf16 --steps 30 --warmup 8 --modes graph 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config'], {k:round(v['stream_ms_median'],3) for k,v in d.items() if isinstance(v,dict)})"; done
  rm -rf $O/p
done
touch $R/dl-dkd_amd/csrc/tower_train.hip; make -C $R/dl-dkd_amd/csrc > /dev/null 2>&1
