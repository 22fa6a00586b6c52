#!/bin/bash
# Round 6 final validation + evidence (one GPU call): full GPU suite, the driver's bench command, rocprofv3 kernel stats and
# PMC traffic of the inference workload (tools/evidence_round6.sh), per-shape C3 / C5 tables + category table, helpers.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06final; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
PWG_BENCH_DETAIL=r06final/bench_detail.json timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench_stderr.log; cat $O/bench_line.json
timeout 600 python tools/profile_train_shapes.py c3 400 > $O/train_shapes_c3.txt 2>&1
timeout 600 python tools/profile_train_shapes.py c5 400 > $O/train_shapes_c5.txt 2>&1
timeout 600 python tools/profile_train_shapes.py c4 400 > $O/train_shapes_c4.txt 2>&1
timeout 600 python tools/profile_train_shapes.py c2 400 > $O/train_shapes_c2.txt 2>&1
python tools/time_by_category.py $O/train_shapes_c3.txt > $O/c3_time_by_category.txt; cat $O/c3_time_by_category.txt
timeout 300 python tools/bench_hbm_helpers.py > $O/hbm_helpers.txt 2>&1
timeout 1500 bash tools/evidence_round6.sh > $O/evidence.log 2>&1; tail -5 $O/evidence.log
