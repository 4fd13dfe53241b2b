#!/bin/bash
# same-box A/B: Charades-shaped training loop (variable caption counts) with the query axis padded to a bucket of 32 / not padded
mkdir -p gpurun_out/r06
out=gpurun_out/r06/ab_train_epoch_c5_query_bucket.txt
: > $out
for rep in 1 2; do
  for b in 32 0; do
    for prec in bf16 mixed; do
      DLDKD_QUERY_BUCKET=$b python3 tools/prof_train_epoch.py 4096 $prec c5 2>/dev/null | grep n_videos >> $out
    done
  done
done
python3 - <<'PY'
import json
for l in open("gpurun_out/r06/ab_train_epoch_c5_query_bucket.txt"):
    d = json.loads(l)
    print(d["config"], d["precision"], "bucket", d["query_bucket"], [round(x, 3) for x in d["ms_per_step_wall"]], "captures", d["captures"], "eager", d["eager_steps"],
          "replays", d["replays"], "prefetched", d["prefetched"], "distinct query counts", len(d["queries_per_batch"]), d["fallbacks"])
PY
