#!/bin/bash
# Round 5: batch folding of the scale discriminators' 1024-channel tail layers (layers/conv.py: _ConvNd._fold_batch).
# Parity of the folded layers, the full-shape C3 / C5 training parity under folding, then the captured C3 / C5 steps
# with and without it.  Result: profiles/r05_fold_batch_ab.txt
mkdir -p gpurun_out/r5fb
O=gpurun_out/r5fb
timeout 300 python -m pytest tests/test_fold_batch_gpu.py -x -q > $O/pytest_fold.txt 2>&1
tail -15 $O/pytest_fold.txt
PWG_FOLD_BATCH=1 timeout 400 python -m pytest tests/test_train_full_shape_gpu.py tests/test_discriminator_gpu.py -x -q -k "c3 or c5 or discriminator" > $O/pytest_full_shape.txt 2>&1
tail -8 $O/pytest_full_shape.txt
{
  for cfg in c3 c5; do
    for rep in 1 2; do
      echo "== $cfg PWG_FOLD_BATCH=0 (rep $rep)"; PWG_FOLD_BATCH=0 timeout 200 python tools/train_replay.py $cfg 26 2>&1 | tail -1
      echo "== $cfg PWG_FOLD_BATCH=1 (rep $rep)"; PWG_FOLD_BATCH=1 timeout 200 python tools/train_replay.py $cfg 26 2>&1 | tail -1
    done
  done
} > $O/steps.txt 2>&1
cat $O/steps.txt
