#!/usr/bin/env python3
"""The keep-partials evaluation at the benchmark's size (16 taxa x 10^6 codon patterns, every internal node's partial stored: 7.2 GB per
evaluation) under the store variants of the per-tree kernel.  usage: PAML_AMD_JIT_STORE=<0|1|2> python tools/keep_probe.py [n_evals]
(2, no stores at all, only with a library built with -DPAML_AMD_JIT_EXPERIMENTS)"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO]
import torch  # noqa: E402
from paml_amd import engine, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
pb = synth.codon_m0_problem(n_tips=16, n_patt=1_000_000, estimate_pi=True)
eng = engine.engine_for(pb, flags=engine.KEEP_PARTIALS)
d = torch.zeros(n + 3, dtype=torch.float64, device="cuda")
eng.set_stream(torch.cuda.current_stream().cuda_stream)
for i in range(3):
    eng.eval_device(pb.tree.branch, d.data_ptr() + 8 * i)
eng.flush(); torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(n):
    eng.eval_device(pb.tree.branch, d.data_ptr() + 8 * (3 + i))
eng.flush(); torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print("PAML_AMD_JIT_STORE=%s kernel %s: %.4f ms per evaluation, lnL %.6f" % (os.environ.get("PAML_AMD_JIT_STORE", "0"), eng.kernel_name, dt * 1e3, float(d[-1])))
