#!/bin/bash
# Round 5: batch folding restricted to layers whose weight PER GROUP is >= 4 MB (the k = 5 tail layer only).
mkdir -p gpurun_out/r5fb
O=gpurun_out/r5fb
timeout 200 python -m pytest tests/test_fold_batch_gpu.py -x -q 2>&1 | tail -2
{
  for cfg in c3 c5; do
    for rep in 1 2 3; do
      echo "== $cfg PWG_FOLD_BATCH=0 (rep $rep)"; PWG_FOLD_BATCH=0 timeout 200 python tools/train_replay.py $cfg 40 2>&1 | tail -1
      echo "== $cfg PWG_FOLD_BATCH=1 per-group rule (rep $rep)"; PWG_FOLD_BATCH=1 timeout 200 python tools/train_replay.py $cfg 40 2>&1 | tail -1
    done
  done
} > $O/steps_rule2.txt 2>&1
cat $O/steps_rule2.txt
