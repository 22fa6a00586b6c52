mkdir -p gpurun_out/r4d
cd /root/repo
for Q in 1 2 3 4; do
  export GPU_MAX_HW_QUEUES=$Q
  echo "GPU_MAX_HW_QUEUES=$Q branches on: $(python tools/train_replay.py c3 16 2>&1 | tail -1)" >> gpurun_out/r4d/queues.txt
done
unset GPU_MAX_HW_QUEUES
for H in 1.0 0.5; do
echo "branch_streams off (one stream, graph) hint $H: $(PWG_NO_BRANCH=1 PWG_CONCURRENCY_HINT=$H python tools/train_replay.py c3 16 2>&1 | tail -1)" >> gpurun_out/r4d/queues.txt
done
echo "branch_streams off, GPU_MAX_HW_QUEUES=1: $(GPU_MAX_HW_QUEUES=1 PWG_NO_BRANCH=1 python tools/train_replay.py c3 16 2>&1 | tail -1)" >> gpurun_out/r4d/queues.txt
echo "eager, branches off: $(PWG_NO_GRAPH=1 PWG_NO_BRANCH=1 python tools/train_replay.py c3 16 2>&1 | tail -1)" >> gpurun_out/r4d/queues.txt
cat gpurun_out/r4d/queues.txt
