#!/bin/bash
# Round-6 rocprofv3 evidence (run on the GPU box; outputs under gpurun_out/prof_r06, summaries copied to profiles/r06_*):
#   1. --kernel-trace --stats of the bench command with eager launches (one dispatch per kernel): inference + training
#   2. the same for inference alone: the conv family's average launch duration there is what bench.py's `roofline`
#      object (HIP events inside the library) must agree with
#   3. PMC passes FETCH_SIZE / WRITE_SIZE (separate runs, only with --kernel-trace): calibration copies with known byte
#      counts in the product kernels' access patterns + the inference workload -> HBM bytes per launch
# bench.py prints a COMPACT line on stdout since round 5; the full record (roofline.kernels ...) is the detail file.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_r06
mkdir -p $O
CMD_FULL="python $R/bench.py --steps 5 --warmup 2 --train-steps 8 --train-warmup 4 --no-cpu-baseline --no-graph --no-extra-configs"
CMD_INF="python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-graph --no-train --no-latency --no-extra-configs"
PWG_BENCH_DETAIL=prof_r06_full_detail.json rocprofv3 --kernel-trace --stats -d $O/trace -o bench -- $CMD_FULL > $O/trace.log 2>&1
python $R/tools/rocprof_summary.py $(ls $O/trace/*/*.db $O/trace/*.db 2>/dev/null | head -1) $O/bench_kernel_stats.csv \
  "rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 2 --train-steps 8 --train-warmup 4 --no-cpu-baseline --no-graph --no-extra-configs (eager launches: every kernel its own dispatch; inference B16 x 800 frames: 2 warm-up + 5 timed + 3 event-profiled forwards + 2 latency shapes; C3 and C5 training: 4 warm-up + 8 timed + 1 event-profiled step each)"
PWG_BENCH_DETAIL=prof_r06_infer_detail.json rocprofv3 --kernel-trace --stats -d $O/trace_infer -o bench -- $CMD_INF > $O/trace_infer.log 2>&1
python $R/tools/rocprof_summary.py $(ls $O/trace_infer/*/*.db $O/trace_infer/*.db 2>/dev/null | head -1) $O/infer_kernel_stats.csv \
  "rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-graph --no-train --no-latency --no-extra-configs (HiFi-GAN V1 inference B16 x 800 frames only: 2 warm-up + 5 timed + 3 event-profiled forwards = 10 x (47 conv + 15 residual-unit + 1 output) launches)"
cp $R/gpurun_out/prof_r06_infer_detail.json $O/infer_bench.json
grep "^{\"metric\"" $O/trace_infer.log | tail -1 > $O/infer_bench_line.json
if [ -z "$NO_PMC" ]; then
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/calib_$C -o p -- $R/tools/probes/pmc_calib_dma.bin 1024 3 > $O/calib_$C.log 2>&1
  PWG_BENCH_DETAIL=prof_r06_pmc_detail.json rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/bench_$C -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graph --no-train --no-latency --no-extra-configs > $O/bench_$C.log 2>&1
done
python - <<PY
import csv, glob, collections, json
out = {}
for run in ("calib", "bench"):
    out[run] = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        agg = collections.defaultdict(lambda: [0, 0.0])
        for f in glob.glob("$O/%s_%s/**/*counter_collection.csv" % (run, c), recursive=True):
            for row in csv.DictReader(open(f)):
                k = row["Kernel_Name"].split("(")[0].replace("void ", "")
                for fam in ("conv1d_mfma_dma_kernel", "resunit_kernel", "dma_copy<4>", "dma_copy<16>"):
                    if fam in k:
                        k = fam
                agg[k][0] += 1; agg[k][1] += float(row["Counter_Value"])
        out[run][c] = {k: {"dispatches": n, "avg_KiB_per_dispatch": v / n} for k, (n, v) in agg.items()}
json.dump(out, open("$O/pmc_raw.json", "w"), indent=1)
for run, d in out.items():
    for c, dd in d.items():
        for k, v in dd.items():
            if any(s in k for s in ("copy", "conv1d_mfma_dma", "resunit_kernel")):
                print(run, c, k, v)
PY
python $R/tools/pmc_round5_summary.py $O/pmc_raw.json $O/infer_bench.json $O/pmc_hbm_traffic.json
fi
ls -la $O | head -30
