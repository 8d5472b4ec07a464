"""Rows of a rocprofv3 --kernel-trace run in start order: start, end, duration (us), kernel, stream, queue.
usage: raw_timeline.py <dir with *kernel_trace.csv> <rows> [first row]"""
import csv, glob, os, sys
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-28:], r.get("Stream_Id", "?"), r.get("Queue_Id", "?")))
rows.sort()
n = int(sys.argv[2])
a = int(sys.argv[3]) if len(sys.argv) > 3 else len(rows) - n
t0 = rows[a][0]
print(len(rows), "kernels")
for r in rows[a:a + n]:
    print("%9.1f %9.1f %8.1f  %-28s s%s q%s" % ((r[0] - t0) / 1e3, (r[1] - t0) / 1e3, (r[1] - r[0]) / 1e3, r[2], r[3], r[4]))
