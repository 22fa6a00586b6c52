#!/bin/bash
# Final validation of round 5: the driver's bench command, per-shape C3 / C5 training profiles, the full GPU suite.
mkdir -p gpurun_out/r5final
O=gpurun_out/r5final
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench_stderr.txt
echo "bench rc=$?"; cp gpurun_out/bench_detail.json $O/bench_detail.json 2>/dev/null
head -c 1500 $O/bench_line.json; echo
PWG_PROF_SHAPES=1 timeout 120 python tools/profile_train_shapes.py c3 400 > $O/train_shapes_c3.txt 2>&1
PWG_PROF_SHAPES=1 timeout 120 python tools/profile_train_shapes.py c5 400 > $O/train_shapes_c5.txt 2>&1
head -3 $O/train_shapes_c3.txt
timeout 420 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1
echo "pytest rc=$?"; tail -5 $O/pytest_gpu.txt
