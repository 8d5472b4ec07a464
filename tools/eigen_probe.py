#!/usr/bin/env python3
"""Time of paml_amd_set_eigen_qrev_batch (61 x 61 codon matrices, one workgroup each) as an optimiser sees it: the call plus the wait
for its result, cold and warm-started after a finite-difference step (1e-6 relative) or a line-search step (5 %), batches of 1 and 3."""
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO]
import torch  # noqa: E402,F401
from paml_amd import engine, models, synth  # noqa: E402


def main():
    rng = np.random.default_rng(1)
    pb = synth.codon_m0_problem(n_tips=6, n_patt=300)
    eng = engine.engine_for(pb)
    pi = pb.pi[0]
    out = {}
    for warm in (0, 1):
        eng.set_eigen_warm_start(warm)
        for nb in (1, 3):
            for name, step in (("fd_1e-6", 1e-6), ("linesearch_5pct", 0.05)):
                kappa, om = 2.0, np.array([0.1, 1.0, 2.5])[:nb]
                ts, sw = [], []
                for it in range(40):
                    kappa *= 1 + step * rng.choice([-1, 1])
                    om = om * (1 + step * rng.choice([-1, 1], size=nb))
                    Qs, mrs = zip(*[models.codon_q(kappa, w, pi) for w in om])
                    Q, P, M = np.array(Qs), np.array([pi] * nb), np.array(mrs)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    eng.set_eigen_qrev_batch(np.arange(nb) + 1, Q, P, M)
                    eng.flush(); torch.cuda.synchronize()
                    ts.append(time.perf_counter() - t0)
                    sw.append(int(eng.eigen_counters()["sweeps"].max()))
                out["%s batch %d %s" % ("warm" if warm else "cold", nb, name)] = dict(ms_median=float(np.median(ts[5:]) * 1e3), sweeps_median=float(np.median(sw[5:])),
                                                                                      sweeps_max=int(max(sw[5:])))
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
