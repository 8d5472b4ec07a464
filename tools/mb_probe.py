import os, sys
sys.path[:0] = ['/root/repo', '/root/repo/tests']
import torch
import helpers
from paml_amd import hostlib
CTL = os.path.join('/root/repo/tests/golden/ctl')
for gname, prog, ctl in (("hiv_m0","codeml","hiv_ns0.ctl"),):
    a = hostlib.Analysis(os.path.join(CTL, ctl), prog)
    r = a.optimize_minb(a.default_x(), verbose=1)
    n_int = a.n_nodes - a.n_tips
    print(gname, r["lnL"], r["n_eval"], r["branch_calls"], r["nodes_recomputed"], r["n_eval"] + r["nodes_recomputed"]/n_int)
