#!/bin/bash
# MFMA-pipe / wait-state counters of the headline inference workload per kernel family (one PMC pass, --kernel-trace only)
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out/pmc_mfma; mkdir -p $O
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/run -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graph --no-train --no-latency --no-extra-configs > $O/run.log 2>&1
python - <<PY
import csv, glob, collections, json
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob("$O/run/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        fam = "conv1d_mfma_dma_kernel" if "conv1d_mfma_dma_kernel" in k else ("resunit_kernel" if "resunit_kernel" in k else None)
        if fam is None:
            continue
        a = agg[fam][row["Counter_Name"]]
        a[0] += 1; a[1] += float(row["Counter_Value"])
out = {}
for fam, d in agg.items():
    e = {c: {"dispatches": n, "sum": v} for c, (n, v) in d.items()}
    def s(c): return d[c][1] if c in d else float("nan")
    e["derived"] = {"mfma_busy_over_sq_busy_x4simd": s("SQ_VALU_MFMA_BUSY_CYCLES") / (4.0 * s("SQ_BUSY_CYCLES")) if "SQ_BUSY_CYCLES" in d else None,
                    "wait_inst_any_over_wave_cycles": s("SQ_WAIT_INST_ANY") / s("SQ_WAVE_CYCLES") if "SQ_WAVE_CYCLES" in d else None,
                    "lds_bank_conflict_over_lds_active": s("SQ_LDS_BANK_CONFLICT") / s("SQ_LDS_IDX_ACTIVE") if "SQ_LDS_IDX_ACTIVE" in d else None}
    out[fam] = e
json.dump(out, open("$O/mfma_counters.json", "w"), indent=1)
print(json.dumps({k: v["derived"] for k, v in out.items()}, indent=1))
PY
