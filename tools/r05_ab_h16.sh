#!/bin/bash
# same-box A/B of the eval towers' 16-bit operand format: fp16 (shipped) vs bf16 (make H16_BF16=1), resident gallery encode (K4b + K5)
# and the K4 / K4b micro-benchmarks.  Leaves the shipped (fp16) build in place.
cd /root/repo
for v in 1 "" 1 ""; do
  rm -f dl-dkd_amd/csrc/build/tower_seq.o dl-dkd_amd/csrc/build/in_proj_h16.o dl-dkd_amd/csrc/build/in_proj_rows128.o dl-dkd_amd/csrc/build/in_proj_rows128b.o dl-dkd_amd/csrc/build/ingest.o
  make -C dl-dkd_amd/csrc -j16 H16_BF16=$v > /dev/null 2>&1
  echo "== operands: $([ -n "$v" ] && echo bf16 || echo fp16)"
  ENC_BATCH=1024 python tools/prof_encode.py resident 2>/dev/null | tail -1
  python tools/bench_k4b.py 3072 time 2>/dev/null | tail -1
done
