#!/bin/bash
# same-box A/B of the query bucket (32 / 16 / 8 / 0): the fixed-shape C5 step (257 queries) and the Charades-shaped loop
mkdir -p gpurun_out/r06
out=gpurun_out/r06/ab_query_bucket_size.txt
: > $out
for b in 32 16 8 0; do
  echo "== DLDKD_QUERY_BUCKET=$b" >> $out
  DLDKD_QUERY_BUCKET=$b python3 tools/bench_train.py --config c5 --prec bf16 --drop 0.15 --modes graph 2>/dev/null | python3 -c "import sys, json; d = json.loads(sys.stdin.read()); print('bench_train c5 bf16', {k: round(v['stream_ms_median'], 3) for k, v in d.items() if isinstance(v, dict) and 'prefetch' not in k})" >> $out
  DLDKD_QUERY_BUCKET=$b python3 tools/prof_train_epoch.py 4096 bf16 c5 2>/dev/null | grep n_videos | python3 -c "import sys, json; d = json.loads(sys.stdin.read()); print('train_epoch c5 bf16', [round(x, 3) for x in d['ms_per_step_wall']], 'captures', d['captures'], 'eager', d['eager_steps'], 'replays', d['replays'])" >> $out
done
cat $out
