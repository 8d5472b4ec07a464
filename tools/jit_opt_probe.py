import json, os, sys, time
sys.path[:0] = [os.environ.get("GRAFT_REPO_ROOT", "/root/repo")]
import torch
from paml_amd import engine, synth
from bench import algorithmic_flops_per_pattern
d_lnl = torch.zeros(1, dtype=torch.float64, device="cuda")
for taxa, n_patt in ((16, 1_000_000), (128, 200_000)):
    pb = synth.codon_m0_problem(n_tips=taxa, n_patt=n_patt)
    eng = engine.engine_for(pb, flags=engine.JIT if hasattr(engine, "JIT") else 0)
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    t0 = time.perf_counter(); eng.eval_device(pb.tree.branch, d_lnl.data_ptr()); eng.flush(); torch.cuda.synchronize(); tf = time.perf_counter() - t0
    for _ in range(2): eng.eval_device(pb.tree.branch, d_lnl.data_ptr())
    eng.flush(); torch.cuda.synchronize(); eng.profile(True)
    for _ in range(5): eng.eval_device(pb.tree.branch, d_lnl.data_ptr())
    eng.flush(); torch.cuda.synchronize(); p = eng.profile_read()
    kms = p["ms_prune"] / p["n_evals"]
    print(os.environ.get("PAML_AMD_JIT_OPT", "-O3"), taxa, eng.kernel_name, "first %.2f s" % tf, "kernel %.3f ms" % kms, "%.1f TF" % (algorithmic_flops_per_pattern(61, taxa) * n_patt / kms / 1e9), float(d_lnl.item()), flush=True)
