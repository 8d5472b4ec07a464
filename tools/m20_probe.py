#!/usr/bin/env python3
"""20-state kernel probe: 32 taxa x 1e5 patterns x 4 classes (random reversible model) and stewart-sized — time per evaluation,
kernel time, algorithmic TFLOP/s, checked against the oracle on a slice.  PAML_AMD_NO_M20=1: the 16x16x4 kernel trimmed to 20
states (round 1)."""
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "tests"), os.path.join(REPO, "oracle")]
import torch  # noqa: E402,F401
import helpers  # noqa: E402
import oracle  # noqa: E402
from paml_amd import engine  # noqa: E402


def run(n_tips, n_patt, K, steps, flags=0):
    pb = helpers.random_problem(20, n_tips, n_patt, K=K, seed=7)
    eng = engine.engine_for(pb, flags=flags)
    br = pb.tree.branch
    for _ in range(3):
        r = eng.eval(br)
    t0 = time.perf_counter()
    for _ in range(steps):
        r = eng.eval(br)
    dt = (time.perf_counter() - t0) / steps
    # back to back, lnL left on the device (the production loop: no host synchronisation between the evaluations, sustained clock)
    d = torch.zeros(256, dtype=torch.float64, device="cuda")
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    for i in range(20):
        eng.eval_device(br, d.data_ptr() + 8 * i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(4 * steps):
        eng.eval_device(br, d.data_ptr() + 8 * (i % 256))
    torch.cuda.synchronize()
    dt_b2b = (time.perf_counter() - t0) / (4 * steps)
    eng.profile(True)
    for _ in range(10):
        eng.eval(br)
    p = eng.profile_read()
    eng.profile(False)
    k = {q: p[q] / p["n_evals"] for q in ("ms_pmat", "ms_prune", "ms_reduce")}
    flops = ((n_tips - 3) * 800 + (2 * n_tips - 3) * 20 + 40) * float(K) * n_patt
    if os.environ.get("M20_NOCHECK"):      # counter runs: nothing but the case's own launches
        eng.close()
        return dict(case="20 states, %d taxa x %d patterns x %d classes" % (n_tips, n_patt, K), ms_per_eval=dt * 1e3, ms_per_eval_b2b=dt_b2b * 1e3, lnL=r["lnL"],
                    tflops=flops / (k["ms_prune"] * 1e-3) / 1e12, **k)
    sub = pb.slice_patterns(0, min(n_patt, 3000))
    ref = oracle.evaluate(sub)
    got = engine.engine_for(sub, flags=engine.JIT).eval(br, want_lnf=True)
    out = dict(case="20 states, %d taxa x %d patterns x %d classes" % (n_tips, n_patt, K), kernel=eng.kernel_name, ms_per_eval=dt * 1e3, ms_per_eval_b2b=dt_b2b * 1e3,
               tflops_b2b_whole_eval=flops / dt_b2b / 1e12,
               tflops=flops / (k["ms_prune"] * 1e-3) / 1e12, frac=flops / (k["ms_prune"] * 1e-3) / 78.6e12, lnL=r["lnL"],
               slice_kernel=None, slice_rel_diff=abs(got["lnL"] - ref["lnL"]) / abs(ref["lnL"]), slice_max_lnf_diff=float(np.max(np.abs(got["lnf"] - ref["lnf"]))), **k)
    eng.close()
    return out


if __name__ == "__main__":
    cases = [(32, 100_000, 4, 20, 0), (16, 1_000_000, 1, 10, 0), (6, 98, 4, 200, engine.JIT)]
    for i, c in enumerate(cases):
        if len(sys.argv) < 2 or int(sys.argv[1]) == i:
            print(json.dumps(run(c[0], c[1], c[2], c[3], flags=c[4])), flush=True)
