#!/bin/bash
# Round 6: spectral-norm power iteration in fewer launches -- parity, then captured C3 / C5 steps old / new alternating
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06sn; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_discriminator_gpu.py tests/test_weight_bank_gpu.py tests/test_conv_ops_gpu.py -q -x > $O/pytest.log 2>&1; tail -2 $O/pytest.log
timeout 900 python -m pytest tests/test_train_full_shape_gpu.py -q -x -k "c3 or c5" > $O/pytest_full.log 2>&1; tail -2 $O/pytest_full.log
for rep in 1 2 3; do
  for cfg in c3 c5; do
    PWG_SN_FUSED=0 timeout 600 python tools/train_replay.py $cfg 60 2>&1 | grep "last 50" | sed "s/^/old $cfg: /" | tee -a $O/replay.txt
    timeout 600 python tools/train_replay.py $cfg 60 2>&1 | grep "last 50" | sed "s/^/new $cfg: /" | tee -a $O/replay.txt
  done
done
