"""Wall time of pamlh_optimize from the control file's initial values, next to the lnL the reference's optimiser reported."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers
from paml_amd import hostlib

CASES = [("hiv_m0", "codeml", "hiv_ns0.ctl"), ("hiv_m1a", "codeml", "hiv_ns1.ctl"), ("hiv_m2a", "codeml", "hiv_ns2.ctl"), ("hiv_m7", "codeml", "hiv_ns7.ctl"),
         ("hiv_m8", "codeml", "hiv_ns8.ctl"), ("hiv_m3", "codeml", "hiv_ns3.ctl"),
         ("lyso_bsa", "codeml", "lyso_bsa.ctl"), ("lyso_bsa_null", "codeml", "lyso_bsa_null.ctl"), ("lyso_bsb", "codeml", "lyso_bsb.ctl"),
         ("ecp_cmc", "codeml", "ecp_cmc.ctl"), ("ecp_cmd", "codeml", "ecp_cmd.ctl"),
         ("horai_mg4", "baseml", "horai_mg4.ctl"), ("lysin_mg4", "codeml", "lysin_mg4.ctl")]
print("# eigen decompositions: %s" % ("on the host (PAMLH_HOST_EIGEN=1)" if os.environ.get("PAMLH_HOST_EIGEN") else "batched on the device (paml_amd_set_eigen_qrev_batch)"))
for gname, prog, ctl in CASES:
    g = helpers.load_golden(gname)
    a = hostlib.Analysis(os.path.join(ROOT, "tests", "golden", "ctl", ctl), prog)
    a.eval_gpu(a.default_x(), want_lnf=False)           # engine creation outside the clock
    t0 = time.perf_counter()
    r = a.optimize(a.default_x())
    dt = time.perf_counter() - t0
    line = "%-14s np %3d  lnL %.6f (reference %.6f)  %5d evaluations  %.3f s" % (gname, a.np, r["lnL"], g.get("mle_lnL", g["lnL"]), r["n_eval"], dt)
    if gname in ("lyso_bsa", "ecp_cmc", "ecp_cmd"):
        t0 = time.perf_counter()
        a.beb_acd(r["x"])
        line += "   BEB %.2f s" % (time.perf_counter() - t0)
    print(line, flush=True)
