#!/bin/bash
# A/B of the XCD-aware tile order of conv1d_mfma_dma_kernel (PWG_DBG=16 = plain grid order): FETCH_SIZE per launch on
# the headline inference workload.  rocprofv3 PMC pass with --kernel-trace only.
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out/pmc_xcd; mkdir -p $O
for MODE in 0 16; do
  PWG_DBG=$MODE rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch_$MODE -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graph --no-train --no-latency --no-extra-configs > $O/fetch_$MODE.log 2>&1
done
python - <<PY
import csv, glob, collections, json
out = {}
for mode in ("0", "16"):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob("$O/fetch_%s/**/*counter_collection.csv" % mode, recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0].replace("void ", "")
            for fam in ("conv1d_mfma_dma_kernel", "resunit_kernel"):
                if fam in k:
                    k = fam
            agg[k][0] += 1; agg[k][1] += float(row["Counter_Value"])
    out["xcd_order" if mode == "0" else "grid_order"] = {k: {"dispatches": n, "FETCH_SIZE_KiB_per_launch_raw": v / n, "read_bytes_per_launch(x2 gfx950)": v / n * 2048} for k, (n, v) in agg.items() if "conv1d" in k or "resunit" in k}
json.dump(out, open("$O/xcd_ab.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
