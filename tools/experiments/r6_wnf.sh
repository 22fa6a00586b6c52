#!/bin/bash
# Round 6: fused slab-sum + weight-norm finisher for short rows, with loads in flight -- micro-benchmark, parity, captured steps old / new
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06wnf; mkdir -p $O
cd $R
PWG_WN_FUSED_SHORT=0 timeout 300 python tools/bench_wgrad_k1.py > $O/bench_old.txt 2>&1; grep -v amdgpu $O/bench_old.txt
timeout 300 python tools/bench_wgrad_k1.py > $O/bench_new.txt 2>&1; grep -v amdgpu $O/bench_new.txt
timeout 900 python -m pytest tests/test_conv_ops_gpu.py tests/test_resstack_gpu.py -q -x > $O/pytest.log 2>&1; tail -2 $O/pytest.log
timeout 900 python -m pytest tests/test_train_full_shape_gpu.py -q -x > $O/pytest_full.log 2>&1; tail -2 $O/pytest_full.log
for rep in 1 2 3; do
  for cfg in c4 c3 c5; do
    PWG_WN_FUSED_SHORT=0 timeout 600 python tools/train_replay.py $cfg 60 2>&1 | grep "last 50" | sed "s/^/old $cfg: /" | tee -a $O/replay.txt
    timeout 600 python tools/train_replay.py $cfg 60 2>&1 | grep "last 50" | sed "s/^/new $cfg: /" | tee -a $O/replay.txt
  done
done
