#!/bin/bash
# Round 5: batch folding on the multi-band MelGAN step (C4, B = 64): off / up to 40 columns per item / up to 64.
mkdir -p gpurun_out/r5fb
O=gpurun_out/r5fb
{
  for rep in 1 2; do
    echo "== c4 PWG_FOLD_BATCH=0 (rep $rep)"; PWG_FOLD_BATCH=0 timeout 200 python tools/train_replay.py c4 26 2>&1 | tail -1
    echo "== c4 PWG_FOLD_BATCH=1 max 40 columns (rep $rep)"; PWG_FOLD_BATCH=1 timeout 200 python tools/train_replay.py c4 26 2>&1 | tail -1
    echo "== c4 PWG_FOLD_BATCH=1 max 64 columns (rep $rep)"; PWG_FOLD_BATCH=1 PWG_FOLD_MAX_COLS=64 timeout 200 python tools/train_replay.py c4 26 2>&1 | tail -1
  done
  echo "== c3 PWG_FOLD_BATCH=1 max 128 columns"; PWG_FOLD_BATCH=1 PWG_FOLD_MAX_COLS=128 timeout 200 python tools/train_replay.py c3 26 2>&1 | tail -1
  echo "== c2 PWG_FOLD_BATCH=0"; PWG_FOLD_BATCH=0 timeout 200 python tools/train_replay.py c2 26 2>&1 | tail -1
  echo "== c2 PWG_FOLD_BATCH=1"; PWG_FOLD_BATCH=1 timeout 200 python tools/train_replay.py c2 26 2>&1 | tail -1
} > $O/steps_c4.txt 2>&1
cat $O/steps_c4.txt
