#!/usr/bin/env python3
"""Average rocprofv3 --pmc counters per kernel from a counter_collection CSV.  usage: pmc_summary.py <dir>"""
import collections, csv, glob, sys
files = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for f in files:
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0][-40:]
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
        cnt[(k, row["Counter_Name"])] += 1
for k in agg:
    print(k, {c: round(v / cnt[(k, c)], 1) for c, v in agg[k].items()})
