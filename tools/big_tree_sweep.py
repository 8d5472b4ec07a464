#!/usr/bin/env python3
"""Per-tree kernel against the interpreter over tree sizes (first compiled build only): where the unrolled kernel stops paying.
usage: python tools/big_tree_sweep.py taxa ..."""
import json, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO]
import torch
from paml_amd import engine, synth
from bench import algorithmic_flops_per_pattern, FP64_PEAK_TFLOPS
os.environ.setdefault("PAML_AMD_JIT_CACHE", "0")
for taxa in [int(a) for a in sys.argv[1:]]:
    pb = synth.codon_m0_problem(n_tips=taxa, n_patt=65_536, seed=taxa)
    eng = engine.engine_for(pb)
    d = torch.zeros(1, dtype=torch.float64, device="cuda")
    eng.set_stream(torch.cuda.current_stream().cuda_stream)

    def rate(n=4):
        eng.eval_device(pb.tree.branch, d.data_ptr()); eng.flush(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            eng.eval_device(pb.tree.branch, d.data_ptr())
        eng.flush(); torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        return ms, algorithmic_flops_per_pattern(61, taxa) * pb.n_patt / (ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS
    t0 = time.perf_counter()
    eng.eval(pb.tree.branch)
    k0, (ms0, f0) = eng.kernel_name, rate()
    while eng.kernel_name == k0 and time.perf_counter() - t0 < float(os.environ.get('BIG_TREE_WAIT_S', '150')):
        time.sleep(0.25)
        eng.eval(pb.tree.branch)
    secs = time.perf_counter() - t0
    ms1, f1 = rate()
    print(json.dumps(dict(taxa=taxa, interpreter=k0, frac_interpreter=round(f0, 3), seconds=round(secs, 1), kernel=eng.kernel_name, ms_per_eval=round(ms1, 3), frac=round(f1, 3))), flush=True)
    eng.close()
