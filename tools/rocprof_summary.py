"""Dump the per-kernel summary (rocprofv3 --kernel-trace --stats, rocpd sqlite output) as CSV.
usage: rocprof_summary.py results.db out.csv [note...]"""
import csv
import sqlite3
import sys

db, out = sys.argv[1], sys.argv[2]
c = sqlite3.connect(db)
rows = list(c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
with open(out, "w", newline="") as f:
    for note in sys.argv[3:]:
        f.write(f"# {note}\n")
    w = csv.writer(f)
    w.writerow(["kernel", "calls", "total_us", "avg_us", "percent"])
    # template instantiations of one kernel family, aggregated (the family is what bench.py's
    # `roofline` object reports: compare its avg_launch_us with the FAMILY row)
    fam = {}
    for name, calls, total, avg, pct in rows:
        base = name.split("<")[0].replace("void ", "").split("(")[0]
        if "<" in name:
            a = fam.setdefault(base, [0, 0.0, 0.0])
            a[0] += calls
            a[1] += total
            a[2] += pct
    for base, (calls, total, pct) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
        w.writerow([f"FAMILY {base}<*>", calls, f"{total:.3f}", f"{total / calls:.3f}", f"{pct:.4f}"])
    for name, calls, total, avg, pct in rows:
        w.writerow([name, calls, f"{total:.3f}", f"{avg:.3f}", f"{pct:.4f}"])
print(f"{len(rows)} kernels -> {out}")
