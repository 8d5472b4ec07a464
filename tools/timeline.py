"""Kernel timeline of consecutive evaluations from a rocprofv3 --kernel-trace run (profiles/r03_timeline.txt): where the batched
P(t) kernel of evaluation i + 1 (side stream) runs relative to the pruning kernel of evaluation i, and what lies between two pruning
kernels on the main stream.  usage: python tools/timeline.py <dir with *kernel_trace.csv> [n_evals_to_print]"""
import csv
import glob
import os
import sys

d = sys.argv[1]
show = int(sys.argv[2]) if len(sys.argv) > 2 else 3
files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
rows = []
for f in files:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:40], r.get("Stream_Id", r.get("Queue_Id", "?"))))
rows.sort()
prune = [i for i, r in enumerate(rows) if r[2].startswith("prune_jit")]
if len(prune) < show + 3:
    raise SystemExit("too few prune_jit launches in the trace (%d)" % len(prune))
# steady state: the last launches of the longest run of equally long prune kernels
med = sorted(rows[i][1] - rows[i][0] for i in prune)[len(prune) // 2]
steady = [i for i in prune if abs((rows[i][1] - rows[i][0]) - med) < 0.1 * med]
gaps, ov = [], []
for a, b in zip(steady, steady[1:]):
    if b - a > 12:
        continue
    gaps.append(rows[b][0] - rows[a][1])
    for k in range(a + 1, b + 1):
        pass
    pm = [rows[k] for k in range(max(0, a - 6), b + 1) if rows[k][2].startswith("pmat") and rows[a][0] <= rows[k][0] <= rows[b][0]]
    for p in pm:
        ov.append((p[0] - rows[a][0], p[1] - rows[a][0], rows[a][1] - rows[a][0]))
print("# %d kernels, %d prune_jit launches, median prune_jit %.1f us" % (len(rows), len(prune), med / 1e3))
if gaps:
    g = sorted(gaps)
    print("# main stream, end of prune_jit(i) -> start of prune_jit(i+1): median %.1f us, min %.1f, max %.1f  (what sits there: reduce_stage1 [+ stage2], launch gaps)"
          % (g[len(g) // 2] / 1e3, g[0] / 1e3, g[-1] / 1e3))
if ov:
    inside = sum(1 for s, e, L in ov if e <= L)
    print("# P(t) kernels starting while a prune_jit runs: %d; finished before that prune_jit ended: %d (%.0f %%); median start at %.0f %% of the prune kernel, median duration %.1f us"
          % (len(ov), inside, 100.0 * inside / len(ov), 100.0 * sorted(s / L for s, e, L in ov)[len(ov) // 2], sorted(e - s for s, e, L in ov)[len(ov) // 2] / 1e3))
last = steady[-(show + 1)]
t0 = rows[last][0]
print("# the last %d evaluations (us from the start of the first prune_jit shown):  start  end  duration  kernel  stream" % show)
for r in rows[last:]:
    print("%10.1f %10.1f %9.1f  %-40s %s" % ((r[0] - t0) / 1e3, (r[1] - t0) / 1e3, (r[1] - r[0]) / 1e3, r[2], r[3]))
