import gc, json, os, sys, time
sys.path[:0] = [os.environ.get("GRAFT_REPO_ROOT", "/root/repo")]
import torch
from paml_amd import engine, synth
d_lnl = torch.zeros(1, dtype=torch.float64, device="cuda")
def point(taxa, n_patt, pre_gc):
    pb = synth.codon_m0_problem(n_tips=taxa, n_patt=n_patt)
    eng = engine.engine_for(pb)
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    for _ in range(3): eng.eval_device(pb.tree.branch, d_lnl.data_ptr())
    torch.cuda.synchronize()
    if pre_gc: gc.collect(); torch.cuda.synchronize()
    ts = []
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(50): eng.eval_device(pb.tree.branch, d_lnl.data_ptr())
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / 50 * 1e3)
    print(taxa, n_patt, "gc" if pre_gc else "  ", ["%.3f" % t for t in ts], flush=True)
    return eng
for pre in (False, True):
    for n in (1000, 10000, 100000):
        e = point(32, n, pre)
