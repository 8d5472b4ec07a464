#!/usr/bin/env python3
"""Several genes (option G) on the per-tree kernels: time per evaluation back to back and the fraction of the FP64 peak of the algorithmic
flops, one gene against G genes of the same data.  usage: python tools/genes_probe.py <4|20> [taxa] [patterns] [genes] [batch]"""
import dataclasses, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO]
import numpy as np
import torch
from paml_amd import engine, synth
from bench import algorithmic_flops_per_pattern, FP64_PEAK_TFLOPS
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
taxa = int(sys.argv[2]) if len(sys.argv) > 2 else 32
npatt = int(sys.argv[3]) if len(sys.argv) > 3 else 100_000
G = int(sys.argv[4]) if len(sys.argv) > 4 else 4
B = int(sys.argv[5]) if len(sys.argv) > 5 else 0
pb1 = synth.nuc_gtr_gamma_problem(n_tips=taxa, n_patt=npatt) if n == 4 else synth.aa_gamma_problem(n_tips=taxa, n_patt=npatt, seed=taxa)
cuts = [npatt * g // G for g in range(G + 1)]
if os.environ.get("GENES_PROBE_EMPTY"):      # all the patterns in the first gene, the others empty
    cuts = [0] + [npatt] * G
pbg = dataclasses.replace(pb1, gene_off=np.array(cuts, dtype=np.int32), gene_rate=np.linspace(0.7, 1.9, G), eigen_of=None, qfactor=None)
d = torch.zeros(64, dtype=torch.float64, device="cuda")
cases = [("1 gene", pb1), ("%d genes" % G, pbg)]
if os.environ.get("GENES_PROBE_ORDER") == "reverse":
    cases = cases[::-1]
if os.environ.get("GENES_PROBE_ORDER") == "twice":
    cases = cases + cases
for name, pb in cases:
    eng = engine.engine_for(pb)
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    for i in range(10):
        eng.eval_device(pb.tree.branch, d.data_ptr() + 8 * i, pb.gene_rate)
    eng.flush(); torch.cuda.synchronize()
    reps = 100
    t0 = time.perf_counter()
    for i in range(reps):
        eng.eval_device(pb.tree.branch, d.data_ptr() + 8 * (i % 64), pb.gene_rate)
    eng.flush(); torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    fl = algorithmic_flops_per_pattern(n, taxa) * pb.K * npatt
    print("%d states, %d taxa x %d patterns x %d classes, %s, kernel %s: %.4f ms per evaluation, %.3f of the FP64 peak, lnL %.6f" %
          (n, taxa, npatt, pb.K, name, eng.kernel_name, ms, fl / (ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS, float(d[0])), flush=True)
    if B:
        br = np.repeat(pb.tree.branch[None, :], B, axis=0) * (1 + 1e-6 * np.arange(B)[:, None])
        gr = np.tile(pb.gene_rate, (B, 1))
        for _ in range(3):
            eng.eval_batch(br, gene_rate=gr)
        t0 = time.perf_counter()
        for _ in range(10):
            v = eng.eval_batch(br, gene_rate=gr)
        msb = (time.perf_counter() - t0) / 10 * 1e3
        eng.profile(True)
        for _ in range(5):
            eng.eval_batch(br, gene_rate=gr)
        pr = eng.profile_read()
        eng.profile(False)
        print("   (stage events, per call: P(t) %.4f ms, pruning %.4f, reduction %.4f)" % tuple(pr[k] / max(1, pr["n_evals"]) for k in ("ms_pmat", "ms_prune", "ms_reduce")))
        import time as _t
        t1 = _t.perf_counter(); eng._L.paml_amd_flush(eng._h); hostonly = _t.perf_counter() - t1
        print("   batch of %d: %.4f ms per call, %.5f per element, %.3f of the FP64 peak (whole call)" % (B, msb, msb / B, fl * B / (msb * 1e-3) / 1e12 / FP64_PEAK_TFLOPS), flush=True)
    eng.close()
