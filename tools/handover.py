"""How long a CU stays empty between a retiring pruning workgroup of evaluation K-1 and the workgroup of evaluation K that
takes it (two pruning streams).  usage: _handover.py <dump> (reads <dump> and <dump>.1 of a PAML_AMD_PROF_TILES run)"""
import sys
import numpy as np


def load(path):
    raw = open(path, "rb").read()
    nb, stride = np.frombuffer(raw[:8], dtype=np.int32)
    nops = stride - 3
    t = np.frombuffer(raw[8 + 4 * nops:], dtype=np.uint64).astype(np.int64).reshape(-1, nb, stride)[0]
    t = t[t[:, 0] > 0]
    start, ends = t[:, 0], t[:, 1:-3]
    ntile = (ends > 0).sum(axis=1)
    last = np.array([ends[b, ntile[b] - 1] for b in range(t.shape[0])])
    first = ends[:, 0]
    return start, last, ntile, first


import os
dumps = [load(f) for f in [sys.argv[1]] + [sys.argv[1] + ".%d" % i for i in (1, 2, 3)] if os.path.exists(f) and os.path.getsize(f) > 8]
dumps = [d for d in dumps if len(d[0])]
dumps.sort(key=lambda d: d[0].min())
a, b = dumps[-2], dumps[-1]
print("%d lanes with a timeline; comparing the last two evaluations" % len(dumps))
us = lambda x: x / 100.0
t0 = a[0].min()
print("earlier evaluation: %d workgroups, start %.1f..%.1f us, end %.1f..%.1f us" % (len(a[0]), 0, us(a[0].max() - t0), us(a[1].min() - t0), us(a[1].max() - t0)))
print("later evaluation:   %d workgroups, start %.1f..%.1f us, end %.1f..%.1f us" % (len(b[0]), us(b[0].min() - t0), us(b[0].max() - t0), us(b[1].min() - t0), us(b[1].max() - t0)))
ea, sb = np.sort(a[1]), np.sort(b[0])
n = min(len(ea), len(sb))
# the later evaluation's first workgroups may sit on CUs that were free all along: align the LAST n starts with the n ends
gap = sb[len(sb) - n:] - ea[len(ea) - n:]
print("hand-over (j-th end of the earlier -> j-th start of the later), us: median %.1f, p10 %.1f, p90 %.1f, min %.1f, max %.1f"
      % tuple(us(f) for f in (np.median(gap), np.percentile(gap, 10), np.percentile(gap, 90), gap.min(), gap.max())))
for name, x in (("earlier", a), ("later", b)):
    d1 = x[3] - x[0]
    per = (x[1] - x[3]) / np.maximum(x[2] - 1, 1)
    print("%s: first tile %.1f us median (p90 %.1f), later tiles %.1f us median; tiles per workgroup %d..%d"
          % (name, us(np.median(d1)), us(np.percentile(d1, 90)), us(np.median(per)), x[2].min(), x[2].max()))
hist = np.histogram(us(gap), bins=[-1e9, 0, 2, 4, 6, 8, 10, 15, 20, 30, 50, 1e9])[0]
print("hand-over histogram (<0, 0-2, 2-4, 4-6, 6-8, 8-10, 10-15, 15-20, 20-30, 30-50, >50 us):", hist.tolist())
