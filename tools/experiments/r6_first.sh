#!/bin/bash
# Round 6, first GPU call: the suite, the bench line, the reference's own training step on the device through stock
# PyTorch-ROCm, a counter pass on the C3 training shapes (item 1) and the per-shape table.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06a; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
PWG_BENCH_DETAIL=r06a/bench_detail.json timeout 900 python bench.py > $O/bench_line.json 2> $O/bench_stderr.log; cat $O/bench_line.json
timeout 1500 python tools/bench_reference_rocm.py $O/reference_rocm_train.txt train c3 c2 c4 > $O/reference_rocm_train.log 2>&1; tail -5 $O/reference_rocm_train.txt
timeout 600 python tools/profile_train_shapes.py c3 400 > $O/train_shapes_c3.txt 2>&1
timeout 900 bash tools/pmc_round6_train.sh c3 > $O/pmc_train.log 2>&1; tail -30 $O/pmc_train.log
