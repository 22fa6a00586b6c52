#!/bin/bash
# Round 6, call f: fresh timeline of the captured C3 step (rocprofv3 kernel trace of graph replays); batch-1 latency
# with / without the single-utterance underfill split (same box).
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06f; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/tl -o p -- python $R/tools/train_replay.py c3 14 > $O/tl.log 2>&1
python $R/tools/graph_timeline.py $(ls $O/tl/*/p_kernel_trace.csv $O/tl/p_kernel_trace.csv 2>/dev/null | head -1) 3 > $O/graph_timeline_c3.txt 2>&1
head -40 $O/graph_timeline_c3.txt
rm -rf $O/tl
cd $R
for rep in 1 2; do
PWG_SPLIT_UNDERFILL=0 timeout 300 python bench.py --no-train --no-cpu-baseline --no-extra-configs --steps 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('old', d['lat_ms'], d['value'])" >> $O/lat.txt
timeout 300 python bench.py --no-train --no-cpu-baseline --no-extra-configs --steps 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('new', d['lat_ms'], d['value'])" >> $O/lat.txt
done
cat $O/lat.txt
timeout 900 python -m pytest tests/test_hifigan_gpu.py tests/test_conv_ops_gpu.py tests/test_streaming_gpu.py tests/test_graphed_inference_gpu.py -q 2>&1 | tail -3
