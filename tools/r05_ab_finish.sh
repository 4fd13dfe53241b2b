#!/bin/bash
# same-box A/B of the finish kernel's workgroup -> tile mapping (FIN_SWZ: XCD-aware 1-D grid vs the plain 2-D grid)
cd /root/repo
for v in 0 1 0 1; do
  rm -f dl-dkd_amd/csrc/build/simpool_eval.o
  make -C dl-dkd_amd/csrc FIN_SWZ=$v > /dev/null 2>&1
  echo "== FIN_SWZ=$v"; python tools/bench_finish.py 2>/dev/null | tail -1
done
rm -f dl-dkd_amd/csrc/build/simpool_eval.o; make -C dl-dkd_amd/csrc > /dev/null 2>&1
