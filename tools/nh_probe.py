import sys, os
sys.path[:0] = ["/root/repo", "/root/repo/tests", "/root/repo/oracle"]
import numpy as np, helpers
from paml_amd import hostlib
n = sys.argv[1] if len(sys.argv) > 1 else "brown_t92_nhomo3_g4"
g = helpers.load_golden(n)
a = hostlib.Analysis("/root/repo/tests/golden/ctl/%s.ctl" % n, "baseml")
r = a.optimize(a.default_x(), max_iter=2000, verbose=True)
print(n, "default:", r["lnL"], r["converged"], r["n_eval"], "ref", g["mle_lnL"])
print("   x", np.round(r["x"], 4))
print("   g", np.round(np.array(g["x"]), 4))
r2 = a.optimize(r["x"], max_iter=2000)
print("again:", r2["lnL"], r2["n_eval"])
r3 = a.optimize(np.array(g["x"]), max_iter=2000)
print("from ref x:", r3["lnL"], r3["n_eval"])
