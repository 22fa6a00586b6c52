#!/bin/bash
# Copy the round-6 evidence written by tools/experiments/final_round6.sh (gpurun_out/r06final, gpurun_out/prof_r06) into
# profiles/ under the round's names.  Run in the repository root after the GPU call has merged its outputs.
set -e
S=gpurun_out/r06final; P=gpurun_out/prof_r06; D=profiles
cp $S/bench_line.json $D/r06_final_bench_line.json
cp $S/bench_detail.json $D/r06_final_bench_detail.json
tail -3 $S/pytest_gpu.log > $D/r06_final_pytest_gpu_tail.txt
for c in c2 c3 c4 c5; do grep -v amdgpu.ids $S/train_shapes_$c.txt > $D/r06_final_train_shapes_$c.txt; done
{ echo "One eager, serial C3 training step (B16 x 8192, tools/profile_train_shapes.py) by layer category (tools/time_by_category.py):"
  echo "left = the final code of round 5 (profiles/r05_zzz_train_shapes_c3.txt), right = the final code of round 6 (profiles/r06_final_train_shapes_c3.txt)."
  echo "Kernel times add up here; the captured step overlaps its eight discriminator branches (47.2 -> 45.7 ms per replayed step)."
  python tools/time_by_category.py profiles/r05_zzz_train_shapes_c3.txt $D/r06_final_train_shapes_c3.txt; } > $D/r06_c3_time_by_category.txt
cp $P/infer_kernel_stats.csv $D/r06_infer_kernel_stats.csv
cp $P/bench_kernel_stats.csv $D/r06_bench_kernel_stats.csv
cp $P/infer_bench_line.json $D/r06_infer_bench_line.json
# The counter summary (and the detail record of ITS run) is committed once: bench.py cites the committed file, so a line
# taken afterwards carries exactly that number.  KEEP_PMC=0 replaces them with this run's.
if [ "${KEEP_PMC:-1}" != "1" ] || [ ! -f $D/r06_pmc_hbm_traffic.json ]; then
  cp $P/infer_bench.json $D/r06_infer_bench_detail.json
  cp $P/pmc_raw.json $D/r06_pmc_hbm_raw.json
  cp $P/pmc_hbm_traffic.json $D/r06_pmc_hbm_traffic.json
fi
ls -la $D | grep r06_ | wc -l
