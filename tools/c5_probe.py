#!/usr/bin/env python3
"""Small-problem latency (BASELINE configs[4]: HIV env, 13 taxa x 79 codon patterns): ms per synchronous evaluation and the
kernels' own times, M0 (K = 1) and M8 (K = 11)."""
import json, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "tests"), os.path.join(REPO, "oracle")]
import numpy as np
import helpers
from paml_amd import engine, hostlib
for g, ctl in (("hiv_m0", "hiv_ns0.ctl"), ("hiv_m8", "hiv_ns8.ctl")):
    gold = helpers.load_golden(g)
    a = hostlib.Analysis(os.path.join(REPO, "tests/golden/ctl", ctl), "codeml")
    pb = a.problem(np.array(gold["x"]))
    for flags in (0, engine.JIT):
        eng = engine.engine_for(pb, flags=flags)
        br = pb.tree.branch
        for _ in range(20):
            r = eng.eval(br)
        t0 = time.perf_counter()
        for _ in range(500):
            r = eng.eval(br)
        dt = (time.perf_counter() - t0) / 500
        eng.profile(True)
        for _ in range(50):
            eng.eval(br)
        p = eng.profile_read()
        eng.profile(False)
        print(json.dumps(dict(case=g, K=pb.K, kernel=eng.kernel_name, ms_per_eval=dt * 1e3, lnL=r["lnL"], ref=gold["lnL"],
                              **{k: p[k] / p["n_evals"] for k in ("ms_pmat", "ms_prune", "ms_reduce")})), flush=True)
        eng.close()
