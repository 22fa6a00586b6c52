#!/bin/bash
# Round 6, call e: dwordx3 LDS-DMA probe; A/B of the round-6 wgrad / planner changes on the captured steps with 50 timed
# steps per run (the 6-step figure of call d was inside the run-to-run spread); forced tile sweep of the C3 categories
# furthest below 100 TFLOP/s.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06e; mkdir -p $O
cd $R
./tools/probes/glds_x3.bin > $O/glds_x3.txt 2>&1; cat $O/glds_x3.txt
OLD="PWG_SPLIT_UNDERFILL=0 PWG_WG_UNDERFILL=0 PWG_WG_FAST23=0 PWG_WG_ROWS=0"
for rep in 1 2 3; do
  for T in c3 c5; do
    env $OLD timeout 300 python tools/train_replay.py $T 60 2>/dev/null | tail -1 | sed "s/^/old /" >> $O/replay.txt
    timeout 300 python tools/train_replay.py $T 60 2>/dev/null | tail -1 | sed "s/^/new /" >> $O/replay.txt
    PWG_SPLIT_UNDERFILL=0 PWG_WG_UNDERFILL=0 timeout 300 python tools/train_replay.py $T 60 2>/dev/null | tail -1 | sed "s/^/new-wgrad-only /" >> $O/replay.txt
  done
done
cat $O/replay.txt
timeout 900 python tools/bench_dsplit.py c3x > $O/c3x_sweep.txt 2>&1; grep -A4 planner $O/c3x_sweep.txt | cut -c1-100
