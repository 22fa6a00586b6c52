#!/bin/bash
# Round 6: eager fork TOGETHER with the data-parallel reducer: N runs of the two-rank graph-vs-eager test, failures counted
# (before ops.slot_add ordered the later slot contributions: 3 of 12; after: 0 of 12, profiles/r06_eager_nan_bisect.txt).
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06ddp; mkdir -p $O
cd $R
fails=0
for i in $(seq 1 ${N:-12}); do
  PWG_EAGER_BRANCH_STREAMS=1 timeout 600 python -m pytest tests/test_ddp_graph_gpu.py -q -x -k segmented_graph_ddp_matches_eager_ddp > $O/run_$i.log 2>&1 || { fails=$((fails+1)); grep -E "AssertionError: \(" $O/run_$i.log | head -2; }
done
echo "forced eager fork + reducer: $fails failing of ${N:-12} runs" | tee $O/summary.txt
timeout 900 python -m pytest tests/test_ddp_graph_gpu.py -q 2>&1 | tail -2
