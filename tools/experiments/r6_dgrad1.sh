#!/bin/bash
# Round 6: first-layer data gradients on the few-output-channel streaming kernels -- parity, per-shape rows, captured steps old / new
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06dg; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_conv_ops_gpu.py tests/test_discriminator_gpu.py tests/test_pwg_melgan_gpu.py -q -x > $O/pytest.log 2>&1; tail -2 $O/pytest.log
timeout 900 python -m pytest tests/test_train_full_shape_gpu.py -q -x > $O/pytest_full.log 2>&1; tail -2 $O/pytest_full.log
for cfg in c4 c3 c2; do
  timeout 600 python tools/profile_train_shapes.py $cfg 400 > $O/shapes_$cfg.txt 2>&1; grep -E "^c[234]|small_cout| M1\(" $O/shapes_$cfg.txt | cut -c1-170
done
for rep in 1 2 3; do
  for cfg in c4 c3 c2; do
    PWG_SMALL_COUT_DGRAD=0 timeout 600 python tools/train_replay.py $cfg 60 2>&1 | grep "last 50" | sed "s/^/old $cfg: /" | tee -a $O/replay.txt
    timeout 600 python tools/train_replay.py $cfg 60 2>&1 | grep "last 50" | sed "s/^/new $cfg: /" | tee -a $O/replay.txt
  done
done
