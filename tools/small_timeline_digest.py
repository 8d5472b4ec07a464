#!/usr/bin/env python3
"""Digest of a rocprofv3 --kernel-trace run of tools/small_timeline.py: the kernels of one evaluation (median duration of each kernel
name, median gap to the previous kernel), for the back-to-back loop and for the read-back loop (told apart by the gap in front of the
evaluation's first kernel).  usage: python tools/small_timeline_digest.py <dir>"""
import csv
import glob
import os
import statistics as st
import sys

rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("paml_amd::", "")[:44]))
rows.sort()
# an evaluation = the kernels from one pmat kernel to the next
starts = [i for i, r in enumerate(rows) if "pmat" in r[2]]
evals = [rows[a:b] for a, b in zip(starts, starts[1:])]
evals = evals[len(evals) // 10:]
if not evals:
    raise SystemExit("no evaluations in the trace")
shape = st.mode(tuple(k[2] for k in e) for e in evals)
evals = [e for e in evals if tuple(k[2] for k in e) == shape]
half = len(evals) // 2
for label, part in (("first half of the run (back to back)", evals[10:half - 10]), ("second half (scalar read back every evaluation)", evals[half + 10:-5])):
    if len(part) < 5:
        continue
    print("# %s: %d evaluations" % (label, len(part)))
    period = st.median(b[0][0] - a[0][0] for a, b in zip(part, part[1:]) if b[0][0] - a[0][0] < 5e6)
    print("#   evaluation period (first kernel to first kernel): %.1f us" % (period / 1e3))
    for j, name in enumerate(shape):
        dur = st.median(e[j][1] - e[j][0] for e in part)
        gap = st.median((e[j][0] - e[j - 1][1]) for e in part) if j else float("nan")
        print("#   %-44s duration %6.1f us   gap in front %6.1f us" % (name, dur / 1e3, gap / 1e3))
