"""Trace of pamlh_optimize on one golden case: python tools/opt_trace.py <golden> <ctl> [scale of the model part of x]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers
from paml_amd import hostlib

g = helpers.load_golden(sys.argv[1])
a = hostlib.Analysis(os.path.join(ROOT, "tests", "golden", "ctl", sys.argv[2]), "codeml")
x0 = np.array(g["x"])
sc = float(sys.argv[3]) if len(sys.argv) > 3 else 1.1
if sc > 0:
    x0[a.ntime:] *= sc
else:
    x0 = a.default_x()
lo, hi = a.bounds()
r = a.optimize(np.clip(x0, lo, hi), verbose=True)
print(r["converged"], r["lnL"], g.get("mle_lnL"), r["n_eval"])
print(np.array2string(r["x"], precision=6, suppress_small=True, max_line_width=200))
print(np.array2string(np.array(g["x"]), precision=6, suppress_small=True, max_line_width=200))
