#!/usr/bin/env python3
"""The branch-local evaluation at the headline size, quick: kernel time (HIP events) of the forming kernel on an internal branch and of the
hit path's polynomial kernel, and their GB/s.  usage: python tools/branch_quick.py [patterns]"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO]
import numpy as np
import torch
from paml_amd import engine, synth
npatt = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
pb = synth.codon_m0_problem(n_tips=16, n_patt=npatt)
t = pb.tree
father = t.father()
internal = [b for b in range(t.n_tips, t.n_nodes) if b != t.root and father[b] >= t.n_tips and father[b] != t.root] or [b for b in range(t.n_tips, t.n_nodes) if b != t.root]
b = internal[0]
eng = engine.engine_for(pb)
ref = eng.eval(t.branch)["lnL"]
def call(ts):
    t0 = time.perf_counter()
    l, dl, ddl = eng.eval_branch(b, np.asarray(ts, dtype=np.float64), t.branch)
    return (time.perf_counter() - t0) * 1e3, l
call([t.branch[b]]); ms, l = call([t.branch[b]])
assert abs(l[0] - ref) <= 1e-11 * abs(ref), (l[0], ref)
eng.profile(True)
hit = []
for i in range(8):
    w, _ = call([t.branch[b] * (1.01 + 0.002 * i)])
    hit.append((w, eng.branch_kernel_ms()))
eng.profile(False)
print("hit path: wall %.4f ms, kernel %.4f ms = %.0f GB/s (%.3f of 8 TB/s)" % (np.mean([h[0] for h in hit[1:]]), np.mean([h[1] for h in hit[1:]]),
      512.0 * npatt / (np.mean([h[1] for h in hit[1:]]) * 1e-3) / 1e9, 512.0 * npatt / (np.mean([h[1] for h in hit[1:]]) * 1e-3) / 8e12))
eng.close()
os.environ["PAML_AMD_NO_COEF_CACHE"] = "1"
eng = engine.engine_for(pb)
eng.eval(t.branch)
call([t.branch[b]])
eng.profile(True)
form = []
for i in range(8):
    w, l = call([t.branch[b]])
    form.append((w, eng.branch_kernel_ms()))
eng.profile(False)
k = np.mean([f[1] for f in form[1:]])
print("forming kernel (internal branch, both partials resident): wall %.4f ms, kernel %.4f ms = %.0f GB/s (%.3f of 8 TB/s), lnL %.6f" %
      (np.mean([f[0] for f in form[1:]]), k, 1536.0 * npatt / (k * 1e-3) / 1e9, 1536.0 * npatt / (k * 1e-3) / 8e12, l[0]))
