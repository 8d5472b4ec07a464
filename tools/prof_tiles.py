#!/usr/bin/env python3
"""Workgroup timeline of the persistent 61-state kernel from a PAML_AMD_PROF_TILES=1 PAML_AMD_PROF_OPS=<dump> run: s_memrealtime
(100 MHz) at workgroup start and at the end of each of its tiles.  usage: prof_tiles.py <dump>"""
import sys
import numpy as np
raw = open(sys.argv[1], "rb").read()
nb, stride = np.frombuffer(raw[:8], dtype=np.int32)
nops = stride - 3
t = np.frombuffer(raw[8 + 4 * nops:], dtype=np.uint64).astype(np.int64).reshape(-1, nb, stride)[0]
t = t[t[:, 0] > 0]      # the persistent grid is smaller than the number of tiles the buffer is sized for
nb = t.shape[0]
start = t[:, 0]
ends = t[:, 1:-3]
entry = t[:, -3]      # kernel entry (kernels with a prologue in front of the first stamp), else 0
core = t[:, -1] - t[:, -2]      # s_memtime (shader clock) between the workgroup's start and the end of its last tile
ntile = (ends > 0).sum(axis=1)
t0 = start.min()
us = lambda x: x / 100.0
last = np.array([ends[b, ntile[b] - 1] for b in range(nb)])
print("workgroups %d, tiles per workgroup %d..%d, span %.1f us" % (nb, ntile.min(), ntile.max(), us(last.max() - t0)))
if entry.any():
    print("prologue (kernel entry -> first stamp): median %.1f us, max %.1f us; entry skew max %.1f us" % (us(np.median(start - entry)), us((start - entry).max()), us(entry.max() - entry.min())))
print("start skew: median %.1f us, max %.1f us" % (us(np.median(start - t0)), us((start - t0).max())))
print("end: first workgroup done at %.1f us, median %.1f, last %.1f" % (us(last.min() - t0), us(np.median(last - t0)), us(last.max() - t0)))
dur = np.diff(np.concatenate([start[:, None], ends], axis=1), axis=1).astype(float)
dur[ends == 0] = np.nan
print("tile duration by round (us): median / p5 / p95")
for r in range(ntile.max()):
    d = dur[:, r][~np.isnan(dur[:, r])]
    print("  round %2d  n=%3d  %.2f / %.2f / %.2f" % (r, d.size, us(np.median(d)), us(np.percentile(d, 5)), us(np.percentile(d, 95))))
busy = np.nansum(dur, axis=1)
print("shader clock while the workgroups ran: median %.0f MHz (min %.0f, max %.0f)" % tuple(f(core / (last - start) * 100.0) for f in (np.median, np.min, np.max)))
print("per-workgroup busy time: median %.1f us, min %.1f, max %.1f" % (us(np.median(busy)), us(busy.min()), us(busy.max())))
per = busy / ntile
print("mean tile time per workgroup: median %.2f us, p5 %.2f, p95 %.2f, max %.2f" % (us(np.median(per)), us(np.percentile(per, 5)), us(np.percentile(per, 95)), us(per.max())))
xcd = np.arange(nb) % 8
for x in range(8):
    print("  XCD %d: mean tile %.2f us, last end %.1f us" % (x, us(np.mean(per[xcd == x])), us((last[xcd == x] - t0).max())))
