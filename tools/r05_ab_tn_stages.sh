# operand ring of gemm_bf16_tn.hip: 2 stages (two workgroups per CU) against 3 (one per CU, two tiles in flight), with split targets
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05tnst
mkdir -p $O
for cfg in "2 384" "3 384" "3 256" "3 224" "2 384" "3 256"; do
  set -- $cfg
  export DLDKD_TN_NST=$1 DLDKD_TN_TARGET=$2
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/p -- python3 $R/tools/bench_train.py --config c3 --prec bf16 --steps 10 --warmup 2 --modes eager > $O/p.log 2>&1
  cd $R
  echo "== NST=$1 target=$2"; python3 tools/kstats.py gpurun_out/r05tnst/p 70 | grep -E "_tn_|dw_finish|inproj_bwd_red" | cut -c1-150
  for c in c3 c5; do python tools/bench_train.py --config $c --prec bf16 --steps 30 --warmup 8 --modes graph 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config'], {k:round(v['stream_ms_median'],3) for k,v in d.items() if isinstance(v,dict)})"; done
  rm -rf $O/p
done
