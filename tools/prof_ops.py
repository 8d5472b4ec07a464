#!/usr/bin/env python3
"""Kernel experiment helper: per-op cycle breakdown of prune_mfma64 from a -DPROF_OPS build
(wave 0 of every workgroup stamps s_memtime at each op).  usage: prof_ops.py <dump> """
import sys
import numpy as np
NAMES = ["INIT_ONES", "INIT_TIP", "MUL_TIP", "PUSH", "MATMUL", "MATMUL_POP", "SCALE", "STORE", "LOAD", "ROOT", "END",
         "SET_TIP", "SET_TIP2", "MUL_TIP2"]
raw = open(sys.argv[1], "rb").read()
nb, stride = np.frombuffer(raw[:8], dtype=np.int32)
nops = stride - 3
codes = np.frombuffer(raw[8:8 + 4 * nops], dtype=np.int32)
tall = np.frombuffer(raw[8 + 4 * nops:], dtype=np.uint64).astype(np.int64)
planes = tall.size // (nb * stride)
tall = tall.reshape(planes, nb, stride)
keep = tall[0][:, 1] != 0          # persistent kernels stamp only the first tile of each resident workgroup
tall = tall[:, keep, :]
nb = int(keep.sum())
t = tall[0]
kstart = t[:, stride - 1]
pro = t[:, 0] - kstart                   # prologue (z staging + first P + barrier)
stamps = t[:, 1:1 + nops]                # time at fetch of op i
dur = np.diff(stamps, axis=1)            # duration of op i (i < nops-1)
total = stamps[:, -1] - kstart
print("blocks %d ops %d   median total %.0f cyc (memtime ticks), prologue %.0f" % (nb, nops, np.median(total), np.median(pro)))
agg = {}
for i in range(nops - 1):
    agg.setdefault(NAMES[codes[i]], []).append(np.median(dur[:, i]))
for k, v in agg.items():
    print("%-12s n=%2d  median/op %8.0f  sum %9.0f  (%.1f%%)" % (k, len(v), np.mean(v), np.sum(v), 100 * np.sum(v) / np.median(total)))
print("per-op medians:", " ".join("%s:%d" % (NAMES[codes[i]][:6], np.median(dur[:, i])) for i in range(nops - 1)))
span = (t[:, 1 + nops - 1].max() - kstart.min())
print("kernel span ticks %d" % span)

if planes >= 3 and tall[1].any():
    s1 = tall[1][:, 1:1 + nops]
    s2 = tall[2][:, 1:1 + nops]
    mm = [i for i in range(nops - 1) if codes[i] in (4, 5)]
    w = np.median(np.stack([s1[:, i] - stamps[:, i] for i in mm]), axis=1)
    m = np.median(np.stack([s2[:, i] - s1[:, i] for i in mm]), axis=1)
    e = np.median(np.stack([stamps[:, i + 1] - s2[:, i] for i in mm]), axis=1)
    print("MATMUL split (median ticks): wait+barrier %s | stage+mfma %s | epilogue+next-fetch %s" % (w.astype(int), m.astype(int), e.astype(int)))
