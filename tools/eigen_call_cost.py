#!/usr/bin/env python3
"""Host-side cost of paml_amd_set_eigen_qrev_batch (the time until the call returns: uploads from pageable memory, the warm start's
bookkeeping, the launch) against the kernel's, by batch size.  usage: python tools/eigen_call_cost.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from paml_amd import engine, models, synth

pb = synth.codon_m0_problem(n_tips=6, n_patt=300)
eng = engine.engine_for(pb)
eng.eval(pb.tree.branch)
pi = pb.pi[0]
eng.set_eigen_warm_start(1)
for nb in (1, 11, 66, 330):
    ret, tot = [], []
    for it in range(12):
        Qs, mrs = zip(*[models.codon_q(2.0 * (1 + 1e-6 * it), 0.1 + 0.01 * k, pi) for k in range(nb)])
        Q, P, M = np.array(Qs), np.array([pi] * nb), np.array(mrs)
        ids = np.arange(nb) + 1
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.set_eigen_qrev_batch(ids, Q, P, M)
        t1 = time.perf_counter()
        eng.flush(); torch.cuda.synchronize()
        t2 = time.perf_counter()
        ret.append(t1 - t0); tot.append(t2 - t0)
    print("batch of %3d matrices: the call returns after %.3f ms, its result is there after %.3f ms" % (nb, np.median(ret[2:]) * 1e3, np.median(tot[2:]) * 1e3), flush=True)
