#!/bin/bash
# same-box A/B of a tower_seq.hip compile-time switch: bash tools/r04_ab_k5.sh TW_FOLD_LN2
cd /root/repo
for v in 0 1 0 1; do
  rm -f dl-dkd_amd/csrc/build/tower_seq.o
  make -C dl-dkd_amd/csrc $1=$v > /dev/null 2>&1
  ENC_BATCH=1024 python tools/prof_encode.py resident 2>/dev/null | tail -1
  python tools/bench_tower.py 1024 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1=$v', {k:round(v[\"ms_median\"],4) for k,v in d.items() if isinstance(v,dict)})"
done
