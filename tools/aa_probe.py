#!/usr/bin/env python3
"""The 20-state model class at scale (codeml seqtype 2 + G4 on synthetic amino-acid patterns): time per evaluation back to back and the
fraction of the FP64 peak of the algorithmic flops.  usage: python tools/aa_probe.py [taxa] [patterns]   (PAML_AMD_M20_HALF / _W12 / PAML_AMD_NO_M20: variants)"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO]
import torch
from paml_amd import engine, synth
from bench import algorithmic_flops_per_pattern, FP64_PEAK_TFLOPS
taxa = int(sys.argv[1]) if len(sys.argv) > 1 else 60
npatt = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
pb = synth.aa_gamma_problem(n_tips=taxa, n_patt=npatt, seed=taxa)
eng = engine.engine_for(pb)
d = torch.zeros(64, dtype=torch.float64, device="cuda")
eng.set_stream(torch.cuda.current_stream().cuda_stream)
for i in range(10):
    eng.eval_device(pb.tree.branch, d.data_ptr() + 8 * i)
eng.flush(); torch.cuda.synchronize()
n = 50
t0 = time.perf_counter()
for i in range(n):
    eng.eval_device(pb.tree.branch, d.data_ptr() + 8 * (i % 64))
eng.flush(); torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / n * 1e3
v = {k: os.environ[k] for k in os.environ if k.startswith("PAML_AMD_M20") or k == "PAML_AMD_NO_M20"}
print("%d taxa x %d patterns x %d classes %s kernel %s: %.4f ms per evaluation, %.3f of the FP64 peak, lnL %.6f" %
      (taxa, npatt, pb.K, v, eng.kernel_name, ms, algorithmic_flops_per_pattern(20, taxa) * pb.K * npatt / (ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS, float(d[0])))
