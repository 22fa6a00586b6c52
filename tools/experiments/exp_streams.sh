mkdir -p gpurun_out/r4g
cd /root/repo
for N in 1 2 3 4 5 8; do
  echo "PWG_MAX_SIDE_STREAMS=$N: $(PWG_MAX_SIDE_STREAMS=$N python tools/train_replay.py c3 16 2>&1 | tail -1)" >> gpurun_out/r4g/streams.txt
done
for N in 2 3; do
  echo "PWG_MAX_SIDE_STREAMS=$N hint 1.0: $(PWG_CONCURRENCY_HINT=1.0 PWG_MAX_SIDE_STREAMS=$N python tools/train_replay.py c3 16 2>&1 | tail -1)" >> gpurun_out/r4g/streams.txt
done
cat gpurun_out/r4g/streams.txt
