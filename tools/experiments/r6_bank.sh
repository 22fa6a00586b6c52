#!/bin/bash
# Round 6: discriminator weight-bank launches beside the generator's forward pass -- parity, then captured steps old / new
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06bank; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_train_full_shape_gpu.py tests/test_weight_bank_gpu.py -q -x > $O/pytest_full.log 2>&1; tail -2 $O/pytest_full.log
for rep in 1 2 3; do
  for cfg in c3 c5 c4; do
    PWG_BANK_BESIDE=0 timeout 600 python tools/train_replay.py $cfg 60 2>&1 | grep "last 50" | sed "s/^/old $cfg: /" | tee -a $O/replay.txt
    timeout 600 python tools/train_replay.py $cfg 60 2>&1 | grep "last 50" | sed "s/^/new $cfg: /" | tee -a $O/replay.txt
  done
done
CFGS="c3" bash tools/experiments/r6_gaps.sh > $O/gaps.txt 2>&1; head -34 $O/gaps.txt
