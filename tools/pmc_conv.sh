#!/bin/bash
# PMC passes on one conv shape (SQ: 8 counters per pass). usage: pmc_conv.sh C K D T B tag
cd /tmp && export TMPDIR=/tmp
R=/root/repo
OUT=$R/gpurun_out/pmc_$6
mkdir -p $OUT
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT/p1 -o p -- python $R/tools/one_conv.py $1 $2 $3 $4 $5 6 > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $OUT/p2 -o p -- python $R/tools/one_conv.py $1 $2 $3 $4 $5 6 > $OUT/p2.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM GRBM_GUI_ACTIVE SQ_CYCLES --kernel-trace --output-format csv -d $OUT/p3 -o p -- python $R/tools/one_conv.py $1 $2 $3 $4 $5 6 > $OUT/p3.log 2>&1
python - <<PY
import csv, glob, collections
for p in ("p1","p2","p3"):
    files = glob.glob("$OUT/%s/**/*counter_collection.csv" % p, recursive=True)
    agg = collections.defaultdict(list)
    for f in files:
        for row in csv.DictReader(open(f)):
            if "conv1d_mfma" in row["Kernel_Name"]:
                agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, v in agg.items():
        print(p, k, "per-dispatch avg %.4g (n=%d)" % (sum(v)/len(v), len(v)))
PY
