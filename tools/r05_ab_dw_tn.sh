# weight-gradient GEMMs of the training step: gemm_bf16_tn.hip (LDS-DMA tiles, transposed LDS reads; DLDKD_DW_TN=1, default) against the
# register-staged kernels of gemm_bf16.hip (DLDKD_DW_TN=0), same box and library, alternating: eager kernel averages under rocprofv3 and
# the replayed step without a profiler
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05dwtn
mkdir -p $O
for rep in 0 1; do for tn in 0 1; do
  export DLDKD_DW_TN=$tn
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/tn_$tn -- python3 $R/tools/bench_train.py --config c3 --prec bf16 --steps 10 --warmup 2 --modes eager > $O/tn_$tn.log 2>&1
  cd $R
  echo "== DLDKD_DW_TN=$tn"; python3 tools/kstats.py gpurun_out/r05dwtn/tn_$tn 60 | grep -E "dw_|_tn_|inproj_bwd|splitk|cast_bf16" | cut -c1-150
  for c in c3 c5; do python tools/bench_train.py --config $c --prec bf16 --steps 30 --warmup 8 --modes graph 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config'], {k:round(v['stream_ms_median'],3) for k,v in d.items() if isinstance(v,dict)})"; done
  rm -rf $O/tn_$tn
done; done
