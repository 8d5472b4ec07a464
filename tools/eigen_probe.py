"""Latency of the batched eigen-decomposition on the device (paml_amd_set_eigen_qrev_batch) by batch size, beside the host path
(models.eigen_rev = LAPACK through numpy, one core) — profiles/r03_eigen.txt."""
import os
import sys
import time

import numpy as np
import torch  # noqa: F401  (before the engine library)

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from paml_amd import engine, models, synth  # noqa: E402

pb = synth.codon_m0_problem(n_tips=6, n_patt=300)
eng = engine.engine_for(pb)
rng = np.random.default_rng(1)
pi = pb.pi[0]
print("# 61 x 61 codon rate matrices; device = upload of Q, pi + Jacobi kernel + stream synchronisation")
for m in (1, 4, 12, 32, 128, 352, 1024):
    Qs, mrs = [], []
    for _ in range(m):
        Q, mr = models.codon_q(float(rng.uniform(1, 5)), float(rng.uniform(0.01, 2)), pi)
        Qs.append(Q); mrs.append(mr)
    Qs = np.array(Qs); pis = np.array([pi] * m); mrs = np.array(mrs)
    ids = np.arange(m) + 1
    for _ in range(2):
        eng.set_eigen_qrev_batch(ids, Qs, pis, mrs)
        torch.cuda.synchronize()
    reps = 10 if m <= 352 else 3
    t0 = time.perf_counter()
    for _ in range(reps):
        eng.set_eigen_qrev_batch(ids, Qs, pis, mrs)
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    sw = eng.eigen_counters()["sweeps"]
    t0 = time.perf_counter()
    for k in range(min(m, 16)):
        models.eigen_rev(Qs[k], pi)
    th = (time.perf_counter() - t0) / min(m, 16)
    print("batch %5d: device %8.3f ms (%.1f us per matrix, sweeps %d..%d)   numpy/LAPACK host %7.3f ms per matrix" % (m, dt * 1e3, dt / m * 1e6, sw.min(), sw.max(), th * 1e3))
