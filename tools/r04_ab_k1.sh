#!/bin/bash
# same-box A/B of a simpool_eval.hip compile-time switch: bash tools/r04_ab_k1.sh K1_GLDS_OFF
cd /root/repo
for v in 0 1 0 1; do
  rm -f dl-dkd_amd/csrc/build/simpool_eval.o
  make -C dl-dkd_amd/csrc $1=$v > /dev/null 2>&1
  echo "$1=$v $(python tools/bench_simpool.py --iters 12 2>/dev/null | tail -2 | tr '\n' ' ')"
done
python -m pytest tests/test_simpool_gpu.py -q -m gpu 2>&1 | tail -1
