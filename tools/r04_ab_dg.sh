#!/bin/bash
# same-box ablation of simpool_bwd_dg_kernel: DG_ABLATE=1 stops after the set-up phases (no gather); kernel averages from rocprofv3
cd /root/repo
R=$PWD
for v in 0 1 0; do
  rm -f dl-dkd_amd/csrc/build/simpool_train.o
  make -C dl-dkd_amd/csrc DG_ABLATE=$v > /dev/null 2>&1
  (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/dgab && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dgab -- python3 $R/tools/bench_train.py --config ${2:-c3} --prec bf16 --steps 12 --warmup 3 --modes eager > /dev/null 2>&1)
  echo "DG_ABLATE=$v $(grep -h 'simpool_bwd_dg' $(find /tmp/dgab -name '*kernel_stats.csv') | cut -d, -f1-4)"
done
