"""Print the kernel timeline (start offset, duration, name) of the tail of a rocprofv3 --kernel-trace CSV: which kernels
overlap which.  Usage: python tools/trace_timeline.py <dir> [last_ms]"""
import csv
import glob
import sys

d = sys.argv[1]
last_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 15.0
files = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60], r.get("Queue_Id", "?")))
rows.sort()
t_end = rows[-1][1]
t0 = t_end - int(last_ms * 1e6)
prev_end = None
for s, e, name, q in rows:
    if s < t0:
        continue
    print("%9.3f ms  +%8.3f ms  q%-4s %s" % ((s - t0) / 1e6, (e - s) / 1e6, q, name))
