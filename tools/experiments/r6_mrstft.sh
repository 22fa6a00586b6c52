#!/bin/bash
# Round 6: MR-STFT loss pairs added as vectors, unpacked once -- parity, then captured C4 / C2 steps
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06mr; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_losses_gpu.py tests/test_pwg_melgan_gpu.py -q -x > $O/pytest.log 2>&1; tail -2 $O/pytest.log
timeout 900 python -m pytest tests/test_train_full_shape_gpu.py -q -x -k "c2 or c4" > $O/pytest_full.log 2>&1; tail -2 $O/pytest_full.log
for rep in 1 2 3; do
  for cfg in c4 c2; do
    timeout 600 python tools/train_replay.py $cfg 60 2>&1 | grep "last 50" | sed "s/^/new $cfg: /" | tee -a $O/replay.txt
  done
done
