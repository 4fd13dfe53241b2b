# operand ring of gemm_bf16_tn.hip (DLDKD_TN_RING = stages * 100 + rows per tile): 264 (shipped), 432, 332, with split targets
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05tnring
mkdir -p $O
for cfg in "264 384" "432 384" "332 384" "332 512" "432 512" "264 384"; do
  set -- $cfg
  export DLDKD_TN_RING=$1 DLDKD_TN_TARGET=$2
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/p -- python3 $R/tools/bench_train.py --config c3 --prec bf16 --steps 10 --warmup 2 --modes eager > $O/p.log 2>&1
  cd $R
  echo "== RING=$1 target=$2"; python3 tools/kstats.py gpurun_out/r05tnring/p 70 | grep -E "_tn_|dw_finish|inproj_bwd_red" | cut -c1-150
  python tools/bench_train.py --config c3 --prec bf16 --steps 30 --warmup 8 --modes graph 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config'], {k:round(v['stream_ms_median'],3) for k,v in d.items() if isinstance(v,dict)})"
  rm -rf $O/p
done
