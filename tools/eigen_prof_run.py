#!/usr/bin/env python3
"""Two cold decompositions of one 61 x 61 codon matrix, then two warm-started ones, each call timed with its wait (the first call of a
process carries the one-time costs); with a library built with -DEIG_PROF=1 (tools/build_variant.sh prof engine_core -DEIG_PROF=1,
PAML_AMD_LIB=...) the kernel prints its s_memtime stamps per phase."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from paml_amd import engine, models, synth

pb = synth.codon_m0_problem(n_tips=6, n_patt=300)
eng = engine.engine_for(pb)
eng.eval(pb.tree.branch)
pi = pb.pi[0]
for i, w in enumerate((0.4, 0.41, 0.42, 0.42000001, 0.43)):
    if i == 2:
        eng.set_eigen_warm_start(1)
    Q, mr = models.codon_q(2.0, w, pi)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.set_eigen_qrev_batch(np.array([1]), np.array([Q]), np.array([pi]), np.array([mr]))
    eng.flush(); torch.cuda.synchronize()
    print("call %d (%s): %.3f ms, sweeps %s" % (i, "warm" if i >= 3 else "cold", (time.perf_counter() - t0) * 1e3, eng.eigen_counters()["sweeps"].tolist()), flush=True)
