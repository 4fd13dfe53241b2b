# split-K target (workgroups per launch) of gemm_bf16_tn.hip's plan: replayed C3 / C5 steps and eager kernel averages
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05dwtn
mkdir -p $O
for tgt in 512 384 768 256 512; do
  export DLDKD_TN_TARGET=$tgt
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/t_$tgt -- python3 $R/tools/bench_train.py --config c3 --prec bf16 --steps 10 --warmup 2 --modes eager > $O/t_$tgt.log 2>&1
  cd $R
  echo "== DLDKD_TN_TARGET=$tgt"; python3 tools/kstats.py gpurun_out/r05dwtn/t_$tgt 60 | grep -E "dw_|_tn_|inproj_bwd_red|splitk" | cut -c1-150
  for c in c3 c5; do python tools/bench_train.py --config $c --prec bf16 --steps 30 --warmup 8 --modes graph 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config'], {k:round(v['stream_ms_median'],3) for k,v in d.items() if isinstance(v,dict)})"; done
  rm -rf $O/t_$tgt
done
