#!/bin/bash
# Round 6: where a batch-1 utterance's replayed forward does not fill the chip
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06gaps; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/tl_b1 -o p -- python $R/tools/infer_replay.py 1 100 20 > $O/tl_b1.log 2>&1; tail -1 $O/tl_b1.log
f=$(ls $O/tl_b1/*/p_kernel_trace.csv $O/tl_b1/p_kernel_trace.csv 2>/dev/null | head -1)
n=$(python - <<PY
import csv
rows=[r["Kernel_Name"] for r in csv.DictReader(open("$f"))]
# launches per replay = distance between the last two launches of the output layer's kernel
idx=[i for i,k in enumerate(rows) if "small_cout" in k]
print(idx[-1]-idx[-2])
PY
)
echo "launches per replay: $n"
python $R/tools/graph_gaps.py $f 64 $n > $O/graph_gaps_b1f100.txt 2>&1; cat $O/graph_gaps_b1f100.txt
rm -rf $O/tl_b1
