"""61-state evaluation on trees beyond the per-tree kernel's 95-tip limit (they run the op-interpreter kernels): TFLOP/s."""
import json, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO]
import torch
from paml_amd import engine, synth
from bench import algorithmic_flops_per_pattern

d_lnl = torch.zeros(1, dtype=torch.float64, device="cuda")
for taxa, n_patt in ((90, 200_000), (96, 200_000), (128, 200_000), (192, 100_000), (192, 400_000)):
    pb = synth.codon_m0_problem(n_tips=taxa, n_patt=n_patt)
    eng = engine.engine_for(pb)
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    t0 = time.perf_counter()
    eng.eval_device(pb.tree.branch, d_lnl.data_ptr())
    eng.flush(); torch.cuda.synchronize()
    t_first = time.perf_counter() - t0       # includes generating and compiling the tree's kernel
    for _ in range(2):
        eng.eval_device(pb.tree.branch, d_lnl.data_ptr())
    eng.flush(); torch.cuda.synchronize()
    eng.profile(True)
    t0 = time.perf_counter()
    for _ in range(5):
        eng.eval_device(pb.tree.branch, d_lnl.data_ptr())
    eng.flush(); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    p = eng.profile_read(); eng.profile(False)
    kms = p["ms_prune"] / max(1, p["n_evals"])
    print(json.dumps(dict(taxa=taxa, n_patt=n_patt, kernel=eng.kernel_name, first_eval_s=t_first, ms_per_eval=dt * 1e3, prune_ms=kms,
                          tflops=algorithmic_flops_per_pattern(61, taxa) * n_patt / (kms * 1e-3) / 1e12)), flush=True)
