#!/usr/bin/env python3
"""Per-tree kernel of a large tree: time per evaluation against the number of patterns (tiles per CU).  usage: python tools/big_tree_npatt.py taxa npatt ..."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO]
os.environ.update(PAML_AMD_JIT="1")
import torch  # noqa: E402
from paml_amd import engine, synth  # noqa: E402
from bench import algorithmic_flops_per_pattern, FP64_PEAK_TFLOPS  # noqa: E402
taxa = int(sys.argv[1])
for npatt in [int(a) for a in sys.argv[2:]]:
    pb = synth.codon_m0_problem(n_tips=taxa, n_patt=npatt, seed=taxa)
    eng = engine.engine_for(pb)
    d = torch.zeros(1, dtype=torch.float64, device="cuda")
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    for _ in range(2):
        eng.eval_device(pb.tree.branch, d.data_ptr())
    eng.flush(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(4):
        eng.eval_device(pb.tree.branch, d.data_ptr())
    eng.flush(); torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 4 * 1e3
    print("%d taxa x %d patterns (%d tiles) %s: %.3f ms per evaluation, %.3f of the FP64 peak" %
          (taxa, npatt, (npatt + 127) // 128, eng.kernel_name, ms, algorithmic_flops_per_pattern(61, taxa) * npatt / (ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS), flush=True)
    eng.close()
