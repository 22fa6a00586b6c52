#!/bin/bash
# Effective shader clock per kernel family = GRBM_GUI_ACTIVE / dispatch wall time (MI355X_MICROARCH.md, "DVFS give-back":
# the chip clocks to its power budget), for the headline inference workload and for eager C3 / C2 training steps.
# rocprofv3 --pmc with --kernel-trace only.  Result: gpurun_out/pmc_clock/clock.json (+ printed table).
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out/pmc_clock; mkdir -p $O
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/infer -o p -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-graph --no-train --no-latency --no-extra-configs > $O/infer.log 2>&1
for T in c3 c2; do
  PWG_NO_GRAPH=1 PWG_NO_BRANCH=1 PWG_WAVENET_WGRAD_STREAM=0 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/$T -o p -- python $R/tools/train_replay.py $T 5 > $O/$T.log 2>&1
done
python - <<PY
import csv, glob, collections, json, re
out = {}
for run in ("infer", "c3", "c2"):
    agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
    for f in glob.glob("$O/%s/**/*counter_collection.csv" % run, recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != "GRBM_GUI_ACTIVE":
                continue
            k = re.sub(r"^void ", "", row["Kernel_Name"]).split("(")[0]
            k = re.sub(r"^pwg::", "", k).split("<")[0]
            dur = float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
            if dur <= 0:
                continue
            a = agg[k]
            a[0] += 1; a[1] += float(row["Counter_Value"]); a[2] += dur
    rows = sorted(agg.items(), key=lambda kv: -kv[1][2])[:14]
    out[run] = {k: {"dispatches": n, "sum_ns": ns, "GRBM_GUI_ACTIVE_per_ns": cyc / ns} for k, (n, cyc, ns) in rows}
    print(run)
    for k, v in out[run].items():
        print(f"  {k:36s} n={v['dispatches']:5d}  {v['sum_ns'] / 1e6:8.2f} ms  cycles/ns = {v['GRBM_GUI_ACTIVE_per_ns']:.3f}")
json.dump(out, open("$O/clock.json", "w"), indent=1)
PY
