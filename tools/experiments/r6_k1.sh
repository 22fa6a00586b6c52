#!/bin/bash
# Round 6: wgrad_k1_kernel -- parity tests, micro-benchmark old / new, captured C4 step old / new (alternating, one box)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06k1; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_conv_ops_gpu.py tests/test_resstack_gpu.py tests/test_optimizers_gpu.py -q -x > $O/pytest.log 2>&1; tail -3 $O/pytest.log
PWG_WG_K1=0 timeout 300 python tools/bench_wgrad_k1.py > $O/bench_old.txt 2>&1; cat $O/bench_old.txt
timeout 300 python tools/bench_wgrad_k1.py > $O/bench_new.txt 2>&1; cat $O/bench_new.txt
for rep in 1 2; do
  for cfg in c4 c2; do
    PWG_WG_K1=0 timeout 600 python tools/train_replay.py $cfg 60 2>&1 | grep "last 50" | sed "s/^/old $cfg: /" | tee -a $O/replay.txt
    timeout 600 python tools/train_replay.py $cfg 60 2>&1 | grep "last 50" | sed "s/^/new $cfg: /" | tee -a $O/replay.txt
  done
done
timeout 900 python -m pytest tests/test_train_full_shape_gpu.py -q -x -k "c4 or c2" > $O/pytest_full.log 2>&1; tail -3 $O/pytest_full.log
timeout 300 python tools/profile_infer_shapes.py 1 100 > $O/infer_shapes_b1f100.txt 2>&1; head -40 $O/infer_shapes_b1f100.txt
