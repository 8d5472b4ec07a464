#!/usr/bin/env python3
"""Timing-only ablation of the 20-state kernel: are the LDS bank conflicts of its tip-row gathers on the critical path?  The same problem
with every tip of every pattern carrying ONE code (all lanes of a gather read the same row: broadcasts, no conflicts) against random tips."""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "tests")]
import torch  # noqa
import helpers
from paml_amd import engine

for const in (False, True, False, True):
    pb = helpers.random_problem(20, 32, 100_000, K=4, seed=7)
    if const:
        pb.z[:] = 3
    eng = engine.engine_for(pb)
    br = pb.tree.branch
    for _ in range(3):
        eng.eval(br)
    eng.profile(True)
    for _ in range(20):
        eng.eval(br)
    p = eng.profile_read()
    eng.profile(False)
    print("%s tips: kernel %s  prune %.4f ms" % ("constant" if const else "random  ", eng.kernel_name, p["ms_prune"] / p["n_evals"]), flush=True)
    eng.close()
