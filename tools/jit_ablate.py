#!/usr/bin/env python3
"""Timing-only probe of the 61-state per-tree kernel under generator switches (PAML_AMD_JIT_* environment): ms per evaluation
of the C4 workload; with --check the lnL is compared with the golden value (ablations that break the results skip it)."""
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "tests")]
import torch  # noqa: E402
from paml_amd import engine, synth  # noqa: E402

n_patt = 1_000_000
K = int(os.environ.get("ABL_K", "1"))
pb = synth.codon_m0_problem(n_tips=16, n_patt=n_patt)
if K > 1:
    import numpy as np
    pb = synth.codon_nssites_problem(pb, 2.0, np.linspace(0.05, 1.5, K), np.full(K, 1.0 / K))
if os.environ.get("ABL_CONST_TIPS"):      # every pattern the same column: the tip-row gathers become broadcasts (no LDS bank conflicts)
    pb.z[:] = pb.z[:, :1]
eng = engine.engine_for(pb)
d = torch.zeros(64, dtype=torch.float64, device="cuda")
eng.set_stream(torch.cuda.current_stream().cuda_stream)
for i in range(10):
    eng.eval_device(pb.tree.branch, d.data_ptr())
eng.flush(); torch.cuda.synchronize()
eng.profile(True)
t0 = time.perf_counter()
for i in range(20):
    eng.eval_device(pb.tree.branch, d.data_ptr() + 8 * i)
eng.flush(); torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 20
p = eng.profile_read()
print(json.dumps(dict(tag=os.environ.get("ABL_TAG", ""), K=K, ms_per_eval=dt * 1e3, kernel_ms=p["ms_prune"] / p["n_evals"], lnL=float(d[0].item()),
                      frac=98637.0 * K * n_patt / (p["ms_prune"] / p["n_evals"] * 1e-3) / 78.6e12)))
