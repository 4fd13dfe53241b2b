# simpool_bwd_dg_kernel: the set-up phases alone (make DG_ABLATE=1: return before the gather) against the whole kernel, C3 bf16 step, eager averages
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05dg
mkdir -p $O
for v in 1 0; do
  touch $R/dl-dkd_amd/csrc/simpool_train.hip
  if [ $v = 1 ]; then make -C $R/dl-dkd_amd/csrc DG_ABLATE=1 > /dev/null 2>&1; else make -C $R/dl-dkd_amd/csrc > /dev/null 2>&1; fi
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/p -- python3 $R/tools/bench_train.py --config c3 --prec bf16 --steps 10 --warmup 2 --modes eager > $O/p.log 2>&1
  cd $R
  echo "== DG_ABLATE=$v"; python3 tools/kstats.py gpurun_out/r05dg/p 70 | grep -E "simpool_bwd" | cut -c1-150
  rm -rf $O/p
done
