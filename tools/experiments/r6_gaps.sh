#!/bin/bash
# Round 6: where the captured C3 / C4 steps do not fill the chip (tools/graph_gaps.py on a rocprofv3 kernel trace)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06gaps; mkdir -p $O
for cfg in ${CFGS:-c3 c4}; do
  rocprofv3 --kernel-trace --output-format csv -d $O/tl_$cfg -o p -- python $R/tools/train_replay.py $cfg 14 > $O/tl_$cfg.log 2>&1
  f=$(ls $O/tl_$cfg/*/p_kernel_trace.csv $O/tl_$cfg/p_kernel_trace.csv 2>/dev/null | head -1)
  python $R/tools/graph_gaps.py $f > $O/graph_gaps_$cfg.txt 2>&1; cat $O/graph_gaps_$cfg.txt
  python $R/tools/graph_timeline.py $f 3 > $O/graph_timeline_$cfg.txt 2>&1
  rm -rf $O/tl_$cfg
done
