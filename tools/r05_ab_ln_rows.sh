# rows per workgroup of the LayerNorm parameter sums in dw_finish_kernel (gemm_bf16.hip): builds with LN_ROWS = 32 / 64 / 128
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05lnrows
mkdir -p $O
for v in 64 32 128 64; do
  touch $R/dl-dkd_amd/csrc/gemm_bf16.hip
  make -C $R/dl-dkd_amd/csrc LN_ROWS=$v > /dev/null 2>&1
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_$v -- python3 $R/tools/bench_train.py --config c3 --prec bf16 --steps 10 --warmup 2 --modes eager > $O/p_$v.log 2>&1
  cd $R
  echo "== LN_ROWS=$v"; python3 tools/kstats.py gpurun_out/r05lnrows/p_$v 70 | grep -E "dw_finish" | cut -c1-150
  for c in c3 c5; do python tools/bench_train.py --config $c --prec bf16 --steps 30 --warmup 8 --modes graph 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config'], {k:round(v['stream_ms_median'],3) for k,v in d.items() if isinstance(v,dict)})"; done
  rm -rf $O/p_$v
done
touch $R/dl-dkd_amd/csrc/gemm_bf16.hip; make -C $R/dl-dkd_amd/csrc > /dev/null 2>&1
