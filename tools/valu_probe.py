"""Kernel probe for the one-pattern-per-lane kernels: BASELINE configs[1] (32 taxa x 1e5 nucleotide patterns, GTR+G4), the same
shape at 4e6 patterns, and a 20-state problem of 32 taxa x 1e5 patterns."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, 'tests')]
import numpy as np
from paml_amd import engine, synth
import helpers
def timed(eng, branch, steps, warmup=3):
    for _ in range(warmup): eng.eval(branch)
    eng.profile(True); t0 = time.perf_counter()
    for _ in range(steps): r = eng.eval(branch)
    dt = (time.perf_counter() - t0) / steps; p = eng.profile_read(); eng.profile(False)
    return dt, p["ms_prune"] / max(1, p["n_evals"]), r["lnL"]
cases = sys.argv[1:] or ["c2", "c2big", "aa20"]
for npatt in [n for c, n in (("c2", 100_000), ("c2big", 4_000_000)) if c in cases]:
    pb = synth.nuc_gtr_gamma_problem(n_tips=32, n_patt=npatt)
    eng = engine.engine_for(pb)
    dt, pr, lnl = timed(eng, pb.tree.branch, 20)
    print("C2 n_patt=%d kernel=%s ms_eval=%.4f prune_ms=%.4f lnL=%.6f GBps=%.0f" % (npatt, eng.kernel_name, dt*1e3, pr, lnl, 7720*npatt/(pr*1e-3)/1e9))
if "aa20" in cases:
  pb = helpers.random_problem(20, 32, 100_000, K=4, seed=7)
  eng = engine.engine_for(pb)
  dt, pr, lnl = timed(eng, pb.tree.branch, 10)
  print("20-state kernel=%s ms_eval=%.4f prune_ms=%.4f lnL=%.6f" % (eng.kernel_name, dt*1e3, pr, lnl))
