#!/usr/bin/env python3
"""Branch-local evaluation (paml_amd_eval_branch = lfuntdd on resident partials) at BASELINE configs[3] size: 16 taxa x 10^6
codon patterns, M0.  Times the first call (every partial formed once), then a minbranches-style walk over the branches
(only the path between consecutive branches is recomputed), with 1 and 4 trial lengths per call; beside it the full-tree
evaluation.  One JSON line."""
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "tests")]
import torch  # noqa: E402,F401
from paml_amd import engine, synth  # noqa: E402


def main():
    n_patt = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    pb = synth.codon_m0_problem(n_tips=16, n_patt=n_patt, estimate_pi=True)
    eng = engine.engine_for(pb)
    t = pb.tree
    for _ in range(3):
        full = eng.eval(t.branch)["lnL"]
    t0 = time.perf_counter()
    for _ in range(5):
        eng.eval(t.branch)
    ms_full = (time.perf_counter() - t0) / 5 * 1e3
    order = []

    def pre(i):
        for c in t.sons[i]:
            order.append(c)
            pre(c)
    pre(t.root)
    b0 = order[0]
    t0 = time.perf_counter()
    l, dl, ddl = eng.eval_branch(b0, np.array([t.branch[b0]]), t.branch)
    ms_first = (time.perf_counter() - t0) * 1e3
    assert abs(l[0] - full) <= 1e-11 * abs(full), (l[0], full)
    c0 = eng.branch_counters()
    res = {}
    for nt in (1, 4):
        times = []
        for b in order:
            ts = t.branch[b] * (1 + 0.05 * np.arange(nt))
            t0 = time.perf_counter()
            l, dl, ddl = eng.eval_branch(b, ts, t.branch)
            times.append((time.perf_counter() - t0) * 1e3)
            assert abs(l[0] - full) <= 1e-11 * abs(full)
        res["walk_nt%d_ms_per_call" % nt] = float(np.mean(times))
        res["walk_nt%d_ms_max" % nt] = float(np.max(times))
    # same branch again: nothing to recompute, the contraction alone
    t0 = time.perf_counter()
    for _ in range(5):
        eng.eval_branch(order[-1], np.array([t.branch[order[-1]]]), t.branch)
    res["same_branch_ms"] = (time.perf_counter() - t0) / 5 * 1e3
    c1 = eng.branch_counters()
    n_int = t.n_nodes - t.n_tips
    out = dict(case="eval_branch, 16 taxa x %d codon patterns, M0" % n_patt, kernel=eng.kernel_name, full_eval_ms=ms_full,
               first_call_ms=ms_first, nodes_first_call=c0["n_nodes"], n_int=n_int,
               nodes_per_call_in_walk=(c1["n_nodes"] - c0["n_nodes"]) / (c1["n_calls"] - c0["n_calls"]),
               partials_resident_GB=61 * 8 * n_patt * n_int / 1e9 * 64 / 61, lnL=full, **res)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
