#!/bin/bash
# same-box A/B of K5's fragment read-ahead depth (TW_DEPTH): resident gallery encode, alternating builds; leaves the shipped build
cd /root/repo
for v in 3 5 3 6; do
  rm -f dl-dkd_amd/csrc/build/tower_seq.o
  make -C dl-dkd_amd/csrc TW_DEPTH=$v > /dev/null 2>&1
  echo "== TW_DEPTH=$v"; ENC_BATCH=1024 python tools/prof_encode.py resident 2>/dev/null | tail -1
done
rm -f dl-dkd_amd/csrc/build/tower_seq.o; make -C dl-dkd_amd/csrc > /dev/null 2>&1
