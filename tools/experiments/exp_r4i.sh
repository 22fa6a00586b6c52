mkdir -p gpurun_out/r4i
cd /root/repo
python -m pytest tests/test_conv_ops_gpu.py tests/test_conv_fuzz_gpu.py tests/test_discriminator_gpu.py -x -q > gpurun_out/r4i/test_conv.log 2>&1; echo "rc=$?" >> gpurun_out/r4i/test_conv.log
export PWG_PAIR_D=0
echo "SPLIT_FILL=0 noprune: $(PWG_SPLIT_FILL=0 PWG_DBG=32 python tools/train_replay.py c3 16 2>&1 | tail -1)" >> gpurun_out/r4i/timing.txt
echo "SPLIT_FILL=0 prune: $(PWG_SPLIT_FILL=0 python tools/train_replay.py c3 16 2>&1 | tail -1)" >> gpurun_out/r4i/timing.txt
echo "SPLIT_FILL=1 prune: $(python tools/train_replay.py c3 16 2>&1 | tail -1)" >> gpurun_out/r4i/timing.txt
echo "SPLIT_FILL=1 prune hint1.0: $(PWG_CONCURRENCY_HINT=1.0 python tools/train_replay.py c3 16 2>&1 | tail -1)" >> gpurun_out/r4i/timing.txt
echo "SPLIT_FILL=1 prune hint0.75: $(PWG_CONCURRENCY_HINT=0.75 python tools/train_replay.py c3 16 2>&1 | tail -1)" >> gpurun_out/r4i/timing.txt
echo "c5 SPLIT_FILL=1: $(python tools/train_replay.py c5 16 2>&1 | tail -1)" >> gpurun_out/r4i/timing.txt
echo "c5 SPLIT_FILL=0: $(PWG_SPLIT_FILL=0 python tools/train_replay.py c5 16 2>&1 | tail -1)" >> gpurun_out/r4i/timing.txt
cat gpurun_out/r4i/timing.txt; tail -n 4 gpurun_out/r4i/test_conv.log
