#!/usr/bin/env python3
"""One search from the control file's initial values to the MLEs (pamlh_optimize) of a golden case, for rocprofv3 --kernel-trace --stats:
which kernels the time of a small-data optimisation goes to.  usage: python tools/mle_profile.py [hiv_m8] [ctl] [prog]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers
from paml_amd import hostlib
gname, ctl, prog = (sys.argv[1:4] + ["hiv_m8", "hiv_ns8.ctl", "codeml"][len(sys.argv) - 1:])[:3]
g = helpers.load_golden(gname)
a = hostlib.Analysis(os.path.join(ROOT, "tests", "golden", "ctl", ctl), prog)
a.eval_gpu(a.default_x(), want_lnf=False)
t0 = time.perf_counter()
r = a.optimize(a.default_x())
print("%s: lnL %.6f (reference %.6f) %d evaluations %.3f s" % (gname, r["lnL"], g.get("mle_lnL", g["lnL"]), r["n_eval"], time.perf_counter() - t0))
