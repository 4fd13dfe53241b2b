# what the Philox4x32 draws cost the training step: the shipped 10 rounds against a diagnostic build with 7 (make PHILOX_ROUNDS=7; every
# mask changes, consistently) - eager kernel averages and replayed steps (run on a scratch copy: the library is left as last built)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05philox
mkdir -p $O
for v in 7 10 7 10; do
  make -C $R/dl-dkd_amd/csrc clean > /dev/null 2>&1
  make -C $R/dl-dkd_amd/csrc -j32 PHILOX_ROUNDS=$v > /dev/null 2>&1
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/p -- python3 $R/tools/bench_train.py --config c3 --prec bf16 --steps 10 --warmup 2 --modes eager > $O/p.log 2>&1
  cd $R
  echo "== PHILOX_ROUNDS=$v"; python3 tools/kstats.py gpurun_out/r05philox/p 70 | grep -E "tt::[fb]|layernorm_kernel<12>|attn_bf16" | cut -c1-150
  for c in c3 c5; do python tools/bench_train.py --config $c --prec bf16 --steps 30 --warmup 8 --modes graph 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config'], {k:round(v['stream_ms_median'],3) for k,v in d.items() if isinstance(v,dict)})"; done
  rm -rf $O/p
done
