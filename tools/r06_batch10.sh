#!/bin/bash
# same-box A/B of the training loop's data path: one-batch-ahead fetch + double-buffered inputs on / off
mkdir -p gpurun_out/r06
out=gpurun_out/r06/ab_train_epoch_prefetch.txt
: > $out
for rep in 1 2; do
  for prec in bf16 mixed; do
    for flag in "" "--no-prefetch"; do
      python3 tools/prof_train_epoch.py 2048 $prec $flag 2>/dev/null | grep n_videos >> $out
    done
  done
done
cat $out
