#!/usr/bin/env python3
"""More than 64 character codes on the per-tree 61-state kernel (JIT_AMB_OVERFLOW): the headline data with 12 ambiguous triplets (73 codes)
against the clean data — ms per evaluation, fraction of the FP64 peak, lnL against the interpreter's.  usage: python tools/amb_probe.py [taxa] [patterns]"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO]
import numpy as np
import torch
from paml_amd import engine, synth
from bench import algorithmic_flops_per_pattern, FP64_PEAK_TFLOPS
taxa = int(sys.argv[1]) if len(sys.argv) > 1 else 16
npatt = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
pb0 = synth.codon_m0_problem(n_tips=taxa, n_patt=npatt)
rates = [float(x) for x in os.environ.get("AMB_RATES", "0.015,0.0005").split(",")]      # missing, each partial code
pba = synth.with_ambiguous_codons(pb0, missing_rate=rates[0], partial_rate=rates[1])
d = torch.zeros(64, dtype=torch.float64, device="cuda")
for name, pb in (("clean, 61 codes", pb0), ("%d codes" % pba.n_codes, pba)):
    eng = engine.engine_for(pb)
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    for i in range(5):
        eng.eval_device(pb.tree.branch, d.data_ptr() + 8 * i)
    eng.flush(); torch.cuda.synchronize()
    reps = 20
    t0 = time.perf_counter()
    for i in range(reps):
        eng.eval_device(pb.tree.branch, d.data_ptr() + 8 * (i % 64))
    eng.flush(); torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    fl = algorithmic_flops_per_pattern(61, taxa) * npatt
    print("%d taxa x %d codon patterns, %s, kernel %s: %.4f ms per evaluation, %.3f of the FP64 peak, lnL %.6f" %
          (taxa, npatt, name, eng.kernel_name, ms, fl / (ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS, float(d[0])), flush=True)
    eng.close()
os.environ["PAML_AMD_JIT"] = "0"
eng = engine.engine_for(pba)
t0 = time.perf_counter()
v = eng.eval(pba.tree.branch)["lnL"]
print("interpreter (%s): lnL %.6f, difference %.3e, %.1f ms" % (eng.kernel_name, v, v - float(d[0]), (time.perf_counter() - t0) * 1e3))
