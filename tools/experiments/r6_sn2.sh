#!/bin/bash
# Round 6: sum_parts with its loads up front, dot_finish folded into the spectral-norm backward kernel: parity + captured steps
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06sn2; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_discriminator_gpu.py tests/test_train_full_shape_gpu.py -q -x -k "not c2 and not c4" > $O/pytest.log 2>&1; tail -2 $O/pytest.log
for rep in 1 2 3; do
  for cfg in c3 c5; do
    timeout 600 python tools/train_replay.py $cfg 60 2>&1 | grep "last 50" | sed "s/^/new $cfg: /" | tee -a $O/replay.txt
  done
done
