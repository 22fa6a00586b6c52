#!/bin/bash
# rocprofv3 --kernel-trace --stats evidence only (the two trace passes of tools/profile_round.sh, no PMC passes)
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp
R=/root/repo
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/trace -o bench -- python $R/bench.py --steps 5 --warmup 2 --train-steps 8 --train-warmup 4 --no-cpu-baseline --no-graph --no-extra-configs > $O/trace.log 2>&1
python $R/tools/rocprof_summary.py $(ls $O/trace/*.db | head -1) $O/kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 2 --train-steps 8 --train-warmup 4 --no-cpu-baseline --no-graph --no-extra-configs (eager launches so that every kernel is a separate dispatch; inference B=16x800 frames: 2 warm-up + 5 timed + 3 event-profiled + 2 latency shapes; training B=16x8192: 4 warm-up + 8 timed + 1 event-profiled steps)"
rocprofv3 --kernel-trace --stats -d $O/trace_infer -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-graph --no-train --no-latency --no-extra-configs > $O/trace_infer.log 2>&1
python $R/tools/rocprof_summary.py $(ls $O/trace_infer/*.db | head -1) $O/infer_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-graph --no-train --no-latency (HiFi-GAN V1 inference B=16x800 frames only: 2 warm-up + 5 timed + 3 event-profiled forwards = 10 x 63 launches)"
grep "^{\"metric\"" $O/trace_infer.log | tail -1 > $O/infer_bench.json
grep "^{\"metric\"" $O/trace.log | tail -1 > $O/bench.json
rm -rf $O/trace $O/trace_infer
