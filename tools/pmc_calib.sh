#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration on a conv with a known byte count in this kernel's own access
# pattern (4-B-per-lane LDS-DMA reads of x, 4-B epilogue loads/stores): C=32 k=3 T=204800 B=16:
#   reads  = x (419.43 MB) + residual addend (419.43 MB) + weights (12 KB);  writes = y (419.43 MB)
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out/pmc_calib; mkdir -p $O
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/$C -o p -- python $R/tools/one_conv.py 32 3 1 204800 16 6 > $O/$C.log 2>&1
done
python - <<PY
import csv, glob
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    v = [float(r["Counter_Value"]) for f in glob.glob("$O/%s/**/*counter_collection.csv" % c, recursive=True) for r in csv.DictReader(open(f)) if "conv1d_mfma" in r["Kernel_Name"]]
    print(c, "dispatches", len(v), "avg KB", sum(v)/len(v), "-> MB", sum(v)/len(v)*1024/1e6)
PY
