#!/usr/bin/env python3
"""Large trees: how long the interpreter serves before the tree's own kernel is compiled (worker thread, JIT_BIG_FLAGS), and the rate
before and after.  usage: python tools/big_tree_compile_probe.py [taxa ...]   (PAML_AMD_JIT_BIG_DEFAULT_FLAGS=1: the compiler's default passes)"""
import json, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO]
import torch
from paml_amd import engine, synth
from bench import algorithmic_flops_per_pattern, FP64_PEAK_TFLOPS

os.environ.setdefault("PAML_AMD_JIT_CACHE", "0")      # (every run compiles)
for taxa in [int(a) for a in sys.argv[1:]] or [96, 192]:
    pb = synth.codon_m0_problem(n_tips=taxa, n_patt=65_536, seed=taxa)
    eng = engine.engine_for(pb)
    d = torch.zeros(1, dtype=torch.float64, device="cuda")
    eng.set_stream(torch.cuda.current_stream().cuda_stream)

    def rate(n=5):
        for _ in range(2):
            eng.eval_device(pb.tree.branch, d.data_ptr())
        eng.flush(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            eng.eval_device(pb.tree.branch, d.data_ptr())
        eng.flush(); torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        return ms, algorithmic_flops_per_pattern(61, taxa) * pb.n_patt / (ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS
    t0 = time.perf_counter()
    v0 = eng.eval(pb.tree.branch)["lnL"]
    k0, (ms0, f0) = eng.kernel_name, rate()
    out = dict(taxa=taxa, first_kernel=k0, ms_per_eval_first=ms0, frac_first=f0)
    for want in ("mfma64_jit_quick", "mfma64_jit"):
        while eng.kernel_name not in (want, "mfma64_jit") and time.perf_counter() - t0 < 180:
            time.sleep(0.25)
            eng.eval(pb.tree.branch)
        if eng.kernel_name == want:
            secs = time.perf_counter() - t0
            ms1, f1 = rate()
            v1 = eng.eval(pb.tree.branch)["lnL"]
            out[want] = dict(seconds=secs, ms_per_eval=ms1, frac=f1, same_lnL_to_1e12=bool(abs(v1 - v0) <= 1e-12 * abs(v0)))
    print(json.dumps(out), flush=True)
    eng.close()
