#!/bin/bash
# Round-3 HBM-traffic evidence (rocprofv3 PMC; FETCH_SIZE and WRITE_SIZE in separate passes, never with a trace domain
# other than --kernel-trace):
#   1. calibration of both counters on known byte counts in the product kernels' access patterns (4-B LDS-DMA,
#      16-B LDS-DMA, 16-B global loads): tools/probes/pmc_calib_dma.bin
#   2. the one-launch WaveNet layer (inference and training form): tools/one_wavenet.py
#   3. the headline inference workload: bench.py (conv family, residual units)
# Raw per-dispatch sums -> gpurun_out/pmc_r03/*.json; tools/pmc_round3_summary.py turns them into profiles/r03_pmc_hbm_traffic.json
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out/pmc_r03; mkdir -p $O
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/calib_$C -o p -- $R/tools/probes/pmc_calib_dma.bin 1024 3 > $O/calib_$C.log 2>&1
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/wn0_$C -o p -- python $R/tools/one_wavenet.py 16 102400 16 0 4 > $O/wn0_$C.log 2>&1
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/wn1_$C -o p -- python $R/tools/one_wavenet.py 6 25600 16 1 6 > $O/wn1_$C.log 2>&1
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/bench_$C -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graph --no-train --no-latency --no-extra-configs > $O/bench_$C.log 2>&1
done
python - <<PY
import csv, glob, collections, json
out = {}
for run in ("calib", "wn0", "wn1", "bench"):
    out[run] = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        agg = collections.defaultdict(lambda: [0, 0.0])
        for f in glob.glob("$O/%s_%s/**/*counter_collection.csv" % (run, c), recursive=True):
            for row in csv.DictReader(open(f)):
                k = row["Kernel_Name"].split("(")[0].replace("void ", "")
                for fam in ("conv1d_mfma_dma_kernel", "resunit_kernel", "wavenet_layer_kernel", "dma_copy<4>", "dma_copy<16>"):
                    if fam in k:
                        k = fam
                agg[k][0] += 1; agg[k][1] += float(row["Counter_Value"])
        out[run][c] = {k: {"dispatches": n, "avg_KiB_per_dispatch": v / n} for k, (n, v) in agg.items()}
json.dump(out, open("$O/pmc_raw.json", "w"), indent=1)
for run, d in out.items():
    for c, dd in d.items():
        for k, v in dd.items():
            if any(s in k for s in ("copy", "wavenet_layer", "conv1d_mfma_dma", "resunit_kernel")):
                print(run, c, k, v)
PY
grep "^{\"metric\"" $O/bench_FETCH_SIZE.log | tail -1 > $O/infer_bench.json
