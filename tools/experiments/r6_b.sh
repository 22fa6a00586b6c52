#!/bin/bash
# Round 6, call b: (1) forced (tile, split-K) sweep of the batch-folded scale-discriminator tail layers; (2) the 30
# generator layers with and without the in-loop pre-activation (what a dual-output epilogue would buy); (3) remaining
# GPU tests after the inference-mode fix.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06b; mkdir -p $O
cd $R
timeout 900 python tools/bench_dsplit.py fold > $O/fold_sweep.txt 2>&1; grep -c planner $O/fold_sweep.txt
timeout 300 python tools/bench_conv.py 16 800 > $O/conv_act1.txt 2>&1; tail -1 $O/conv_act1.txt
PWG_BENCH_ACT=0 timeout 300 python tools/bench_conv.py 16 800 > $O/conv_act0.txt 2>&1; tail -1 $O/conv_act0.txt
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
