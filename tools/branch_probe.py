#!/usr/bin/env python3
"""Branch-local evaluation (paml_amd_eval_branch = lfuntdd on resident partials) at BASELINE configs[3] size: 16 taxa x 10^6
codon patterns, M0.  Wall time per call (one host synchronisation each), as minbranches issues them:
  first_call           every partial formed once
  walk_form_nt1        moving to the next branch of the pre-order walk: the re-oriented node(s) re-formed + the contraction (coefficients formed)
  walk_hit_nt4         four further trial lengths on that branch (served from the stored coefficients)
  same_branch_form     the contraction alone, coefficients formed again on the same branch (PAML_AMD_NO_COEF_CACHE=1 engine), nt = 1 and 4
  same_branch_hit      a further trial length on the same branch
beside the full-tree evaluation.  One JSON line.  PAML_AMD_NO_BRANCH_EIG=1: round 2's P / dP / ddP kernels."""
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "tests")]
import torch  # noqa: E402,F401
from paml_amd import engine, synth  # noqa: E402


def timed(f, reps=1):
    t0 = time.perf_counter()
    for _ in range(reps):
        r = f()
    return (time.perf_counter() - t0) / reps * 1e3, r


def main():
    n_patt = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    nocheck = bool(os.environ.get("PROBE_NOCHECK"))      # timing ablations return garbage
    pb = synth.codon_m0_problem(n_tips=16, n_patt=n_patt, estimate_pi=True)
    eng = engine.engine_for(pb)
    t = pb.tree
    for _ in range(3):
        full = eng.eval(t.branch)["lnL"]
    ms_full, _ = timed(lambda: eng.eval(t.branch), 5)
    order = []

    def pre(i):
        for c in t.sons[i]:
            order.append(c)
            pre(c)
    pre(t.root)
    b0 = order[0]
    ms_first, (l, dl, ddl) = timed(lambda: eng.eval_branch(b0, np.array([t.branch[b0]]), t.branch))
    assert nocheck or abs(l[0] - full) <= 1e-11 * abs(full), (l[0], full)
    c0 = eng.branch_counters()
    form, hit = [], []
    for cycle in range(2):
        for b in order:
            ms, (l, dl, ddl) = timed(lambda: eng.eval_branch(b, np.array([t.branch[b]]), t.branch))
            form.append(ms)
            assert nocheck or abs(l[0] - full) <= 1e-11 * abs(full)
            ts = t.branch[b] * (1 + 0.05 * np.arange(1, 5))
            ms, _ = timed(lambda: eng.eval_branch(b, ts, t.branch))
            hit.append(ms)
    c1 = eng.branch_counters()
    res = dict(walk_form_nt1_ms=float(np.mean(form)), walk_form_nt1_ms_max=float(np.max(form)), walk_hit_nt4_ms=float(np.mean(hit)))
    b = [v for v in order if v >= t.n_tips][-1]      # an internal branch: both ends' partials are read
    res["same_branch_hit_nt1_ms"], _ = timed(lambda: eng.eval_branch(b, np.array([t.branch[b] * 1.01]), t.branch), 10)
    res["same_branch_hit_nt4_ms"], _ = timed(lambda: eng.eval_branch(b, t.branch[b] * (1 + 0.05 * np.arange(4)), t.branch), 10)
    eng.close()
    os.environ["PAML_AMD_NO_COEF_CACHE"] = "1"
    eng = engine.engine_for(pb)
    del os.environ["PAML_AMD_NO_COEF_CACHE"]
    eng.eval_branch(b, np.array([t.branch[b]]), t.branch)
    res["same_branch_form_nt1_ms"], (l, _, _) = timed(lambda: eng.eval_branch(b, np.array([t.branch[b]]), t.branch), 10)
    assert nocheck or abs(l[0] - full) <= 1e-11 * abs(full)
    res["same_branch_form_nt4_ms"], _ = timed(lambda: eng.eval_branch(b, t.branch[b] * (1 + 0.05 * np.arange(4)), t.branch), 10)
    tipb = 3
    eng.eval_branch(tipb, np.array([t.branch[tipb]]), t.branch)
    res["same_tip_branch_form_nt1_ms"], _ = timed(lambda: eng.eval_branch(tipb, np.array([t.branch[tipb]]), t.branch), 10)
    n_int = t.n_nodes - t.n_tips
    out = dict(case="eval_branch, 16 taxa x %d codon patterns, M0" % n_patt, kernel=eng.kernel_name, full_eval_ms=ms_full,
               first_call_ms=ms_first, nodes_first_call=c0["n_nodes"], n_int=n_int,
               nodes_per_forming_call_in_walk=(c1["n_nodes"] - c0["n_nodes"]) / len(form), coef_hits=c1["coef_hits"],
               partials_resident_GB=64 * 8 * n_patt * n_int / 1e9, coefficients_GB=64 * 8 * n_patt / 1e9, lnL=full, **res)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
