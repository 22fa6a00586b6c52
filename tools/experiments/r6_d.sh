#!/bin/bash
# Round 6, call c: A/B on ONE box of the round-6 planner / weight-gradient changes (underfill split-K of the conv kernel,
# underfill slices + activation-free / compile-time-stride loops of the strided weight gradients) against the round-5
# behaviour (env switches), per shape (eager, serial) and on the captured steps.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06d; mkdir -p $O
cd $R
OLD="PWG_SPLIT_UNDERFILL=0 PWG_WG_UNDERFILL=0 PWG_WG_FAST23=0 PWG_WG_ROWS=0"
timeout 600 python -m pytest tests/test_conv_ops_gpu.py tests/test_conv_fuzz_gpu.py tests/test_fold_batch_gpu.py tests/test_discriminator_gpu.py tests/test_train_full_shape_gpu.py -q > $O/pytest_conv.log 2>&1; tail -3 $O/pytest_conv.log
env $OLD timeout 600 python tools/profile_train_shapes.py c3 400 > $O/shapes_c3_old.txt 2>&1
timeout 600 python tools/profile_train_shapes.py c3 400 > $O/shapes_c3_new.txt 2>&1
for T in c3 c5 c2 c4; do
  for rep in 1 2; do
    env $OLD timeout 300 python tools/train_replay.py $T 40 2>/dev/null | tail -1 | sed "s/^/old /" >> $O/replay.txt
    timeout 300 python tools/train_replay.py $T 40 2>/dev/null | tail -1 | sed "s/^/new /" >> $O/replay.txt
  done
done
cat $O/replay.txt
