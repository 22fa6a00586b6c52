mkdir -p gpurun_out/r4c
cd /root/repo
for Q in default 8 16; do
  if [ $Q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$Q; fi
  for H in 0.5 1.0 0.25; do
    echo "GPU_MAX_HW_QUEUES=$Q PWG_CONCURRENCY_HINT=$H: $(PWG_CONCURRENCY_HINT=$H python tools/train_replay.py c3 16 2>&1 | tail -1)" >> gpurun_out/r4c/queues.txt
  done
done
unset GPU_MAX_HW_QUEUES
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o p -- python /root/repo/tools/train_replay.py c3 10 > /dev/null 2>&1
F=$(find /tmp/tl -name "*kernel_trace.csv" | head -1)
python - $F <<'PY'
import csv, sys
rows=[]
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"], r["Stream_Id"], r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"], r["Workgroup_Size_X"]))
rows.sort()
opt=[i for i,r in enumerate(rows) if "adam_multi" in r[2]]
# last step = between opt[-3] and opt[-1]
lo, hi = opt[-3]+1, opt[-1]+1
t0=rows[lo][0]
with open("/root/repo/gpurun_out/r4c/last_step_trace.csv","w") as f:
    f.write("start_us,dur_us,queue,stream,wgs,kernel\n")
    for s,e,n,q,st,gx,gy,gz,wx in rows[lo:hi]:
        wgs=(int(gx)//max(int(wx),1))*int(gy)*int(gz)
        n=n.replace("void pwg::","").replace("pwg::","")[:90].replace(",",";")
        f.write(f"{(s-t0)/1e3:.1f},{(e-s)/1e3:.1f},{q},{st},{wgs},{n}\n")
PY
cat /root/repo/gpurun_out/r4c/queues.txt
