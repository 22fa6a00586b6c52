#!/bin/bash
# Round 6: wgrad_k1_kernel with >= 256 workgroups + the 16-B single-input-channel weight gradient: parity, micro-benchmark,
# per-shape tables of C4 / C3 (eager, serial), captured steps
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06k1b; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_conv_ops_gpu.py tests/test_resstack_gpu.py -q -x > $O/pytest.log 2>&1; tail -2 $O/pytest.log
timeout 300 python tools/bench_wgrad_k1.py > $O/bench_new.txt 2>&1; cat $O/bench_new.txt
timeout 600 python tools/profile_train_shapes.py c4 400 > $O/shapes_c4.txt 2>&1; grep -E "^c4|small_cin_wgrad|wgrad_k1|adam" $O/shapes_c4.txt
timeout 600 python tools/profile_train_shapes.py c3 400 > $O/shapes_c3.txt 2>&1; grep -E "^c3|small_cin_wgrad|adam" $O/shapes_c3.txt
for cfg in c4 c3 c5 c2; do
  timeout 600 python tools/train_replay.py $cfg 60 2>&1 | grep "last 50" | sed "s/^/new $cfg: /" | tee -a $O/replay.txt
done
