#!/usr/bin/env python3
"""bench.py's c2_large workload alone (32 taxa x 10^7 nucleotide patterns, GTR + Gamma-4: the 10^6-pattern alignment ten times over) — for
rocprofv3 kernel statistics and FETCH_SIZE / WRITE_SIZE passes (tools/collect_profiles_r06.sh).  usage: python tools/c2large_probe.py [evaluations]"""
import dataclasses, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO]
import numpy as np
import torch
from paml_amd import engine, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
one = synth.nuc_gtr_gamma_problem(n_tips=32, n_patt=1_000_000)
pb = dataclasses.replace(one, z=np.ascontiguousarray(np.tile(one.z, (1, 10))), weights=np.tile(one.weights, 10), gene_off=None, eigen_of=None, qfactor=None)
eng = engine.engine_for(pb)
d = torch.zeros(64, dtype=torch.float64, device="cuda")
eng.set_stream(torch.cuda.current_stream().cuda_stream)
for i in range(3):
    eng.eval_device(pb.tree.branch, d.data_ptr())
eng.flush(); torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(n):
    eng.eval_device(pb.tree.branch, d.data_ptr() + 8 * (i % 64))
eng.flush(); torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / n * 1e3
print("32 taxa x %d patterns x 4 classes, kernel %s: %.4f ms per evaluation, lnL %.6f" % (pb.n_patt, eng.kernel_name, ms, float(d[0])))
