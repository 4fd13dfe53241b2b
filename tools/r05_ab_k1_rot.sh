#!/bin/bash
# same-box A/B of the clock-rotated query-tile order of K1 (K1_ROT): kernel time (HIP events) and the
# FETCH_SIZE / WRITE_SIZE counters (one --pmc pass each) per launch
R=/root/repo; O=$R/gpurun_out/r05rot; mkdir -p $O
cd $R
for v in 0 1 2 3 0 1; do
  rm -f dl-dkd_amd/csrc/build/simpool_eval.o
  make -C dl-dkd_amd/csrc K1_ROT=$v > /dev/null 2>&1
  echo "== K1_ROT=$v"; python tools/bench_simpool.py --iters 10 2>/dev/null | tail -2
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/f_$v -- python3 $R/tools/bench_simpool.py --iters 3 > $O/f_$v.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/w_$v -- python3 $R/tools/bench_simpool.py --iters 3 > $O/w_$v.log 2>&1
  cd $R
  python3 tools/pmc_summary.py simpool_eval16p_kernel 17.0 $O/s_$v.json $O/f_$v $O/w_$v > /dev/null 2>&1; python3 -c "import json; d=json.load(open('$O/s_$v.json'))['derived']; print({k: round(v,3) for k,v in d.items() if 'hbm' in k})"
  rm -rf $O/f_$v $O/w_$v
done
rm -f dl-dkd_amd/csrc/build/simpool_eval.o; make -C dl-dkd_amd/csrc > /dev/null 2>&1
