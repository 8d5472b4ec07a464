"""Back-to-back time per evaluation of a sequence of engines in one process, e.g. 8p,8c,4p,1p: N = the shard divisor of the headline
workload (10^6 / N codon patterns, 16 taxa, M0), p plain / c inside a one-rank RCCL communicator.  Every evaluation of a run has its
own result slot; all must be equal.  Used with PAML_AMD_DUAL / PAML_AMD_LANES / PAML_AMD_COMM_CUS for profiles/r03_dual_stream.txt."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paml_amd import distributed, engine, synth
full = synth.codon_m0_problem(n_tips=16, n_patt=1_000_000, estimate_pi=True)
out = []
for spec in sys.argv[1].split(","):
    N, comm = int(spec[:-1]), spec[-1] == "c"
    lo, hi = distributed.shard_bounds(full.n_patt, N, 0)
    pb = full.slice_patterns(lo, hi) if N > 1 else full
    eng = engine.engine_for(pb)
    eng.comm_init(0, 1, engine.comm_unique_id() if comm else None, pb.n_patt, 0)
    d = torch.zeros(400, dtype=torch.float64, device="cuda")
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    for i in range(10):
        eng.eval_device(pb.tree.branch, d.data_ptr() + 8 * i)
    eng.flush(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(200):
        eng.eval_device(pb.tree.branch, d.data_ptr() + 8 * (10 + i))
    eng.flush(); torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 200 * 1e3
    vals = d[:210].cpu().numpy()
    out.append("%s %.4f%s" % (spec, ms, "" if (vals == vals[0]).all() else " MISMATCH"))
    eng.close()
print("  ".join(out))
