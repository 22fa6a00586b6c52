#!/bin/bash
# rocprofv3 evidence for the round: kernel-trace stats of the default bench command, then separate
# PMC passes (FETCH_SIZE / WRITE_SIZE cannot share a pass: TCC has 4 slots, MI355X_MICROARCH.md).
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp
R=/root/repo
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
if [ -z "$PMC_ONLY" ]; then
rocprofv3 --kernel-trace --stats -d $O/trace -o bench -- python $R/bench.py --steps 5 --warmup 2 --train-steps 8 --train-warmup 4 --no-cpu-baseline --no-graph --no-extra-configs > $O/trace.log 2>&1
fi
[ -z "$PMC_ONLY" ] && python $R/tools/rocprof_summary.py $(ls $O/trace/*.db | head -1) $O/kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 2 --train-steps 8 --train-warmup 4 --no-cpu-baseline --no-graph --no-extra-configs (eager launches so that every kernel is a separate dispatch; inference B=16x800 frames: 2 warm-up + 5 timed + 3 event-profiled + 2 latency shapes; training B=16x8192: 4 warm-up + 8 timed + 1 event-profiled steps)"
if [ -z "$PMC_ONLY" ]; then
# inference-only trace: the conv family's average launch duration here is the one bench.py's `roofline`
# object must agree with (same workload, eager launches = one dispatch per kernel)
rocprofv3 --kernel-trace --stats -d $O/trace_infer -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-graph --no-train --no-latency --no-extra-configs > $O/trace_infer.log 2>&1
python $R/tools/rocprof_summary.py $(ls $O/trace_infer/*.db | head -1) $O/infer_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-graph --no-train --no-latency (HiFi-GAN V1 inference B=16x800 frames only: 2 warm-up + 5 timed + 3 event-profiled forwards = 10 x 78 conv launches)"
grep "^{\"metric\"" $O/trace_infer.log | tail -1 > $O/infer_bench.json
fi
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_$C -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graph --no-train --no-latency --no-extra-configs > $O/pmc_$C.log 2>&1
done
python - <<PY
import csv, glob, collections, json
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob("$O/pmc_%s/**/*counter_collection.csv" % c, recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0]
            k = "conv1d_mfma_dma_kernel" if "conv1d_mfma_dma_kernel" in k else k
            k = "resunit_kernel" if "resunit_kernel" in k else k
            agg[k][0] += 1; agg[k][1] += float(row["Counter_Value"])
    out[c] = {k: {"dispatches": n, "sum": v, "avg_per_dispatch": v / n} for k, (n, v) in agg.items() if "pwg" in k or "conv1d" in k or "resunit" in k}
json.dump(out, open("$O/pmc_hbm.json", "w"), indent=1)
for c, d in out.items():
    for k, v in d.items(): print(c, k, v)
PY

# ---- training shapes (C3 step, eager): HBM traffic and LDS / MFMA counters per kernel family
if [ -n "$TRAIN_PMC" ]; then
for PASS in "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY"; do
  TAGP=$(echo $PASS | cut -d" " -f1)
  rocprofv3 --pmc $PASS --kernel-trace --output-format csv -d $O/train_pmc_$TAGP -o p -- python $R/bench.py --steps 1 --warmup 1 --train-steps 2 --train-warmup 2 --no-cpu-baseline --no-graph --no-latency --no-extra-configs > $O/train_pmc_$TAGP.log 2>&1
done
python - <<PY
import csv, glob, collections, json
out = {}
for f in glob.glob("$O/train_pmc_*/**/*counter_collection.csv", recursive=True):
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "")
        for fam in ("conv1d_mfma_dma_kernel", "conv1d_wgrad_kernel", "conv1d_mfma_kernel", "resunit_kernel"):
            if fam in k:
                k = fam
        if "pwg" not in row["Kernel_Name"]:
            continue
        a = agg[k][row["Counter_Name"]]
        a[0] += 1; a[1] += float(row["Counter_Value"])
    for k, d in agg.items():
        for c, (n, v) in d.items():
            out.setdefault(k, {})[c] = {"dispatches": n, "sum": v, "avg_per_dispatch": v / n}
json.dump(out, open("$O/train_pmc.json", "w"), indent=1)
print("train_pmc.json:", len(out), "kernel families")
PY
fi
