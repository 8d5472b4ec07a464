"""Stress of the eval_device fast path: many short sequences of queued evaluations with interleaved entry points, checked
against in-order evaluations.  python tools/pipe_stress.py [rounds] [big]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch
import helpers
from paml_amd.engine import engine_for

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 30
big = len(sys.argv) > 2 and sys.argv[2] == "big"      # large enough for the two pruning streams; 61 / 20 states, every third round in a one-rank communicator
bad = 0
for r in range(rounds):
    if big:
        pb = helpers.random_problem((61, 20)[r % 2], 8 + r % 5, 52000 + 1013 * r, K=2, seed=123 + r)
    else:
        pb = helpers.random_problem(61, 12, 3000 + 97 * r, K=2, seed=123 + r)
    eng = engine_for(pb)
    if big and r % 3 == 2:
        from paml_amd import engine as E
        eng.comm_init(0, 1, E.comm_unique_id(), pb.n_patt, 0)
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    rng = np.random.default_rng(r)
    brs = [pb.tree.branch * rng.uniform(0.5, 1.5, pb.tree.n_nodes) for _ in range(16)]
    want = np.array([eng.eval(b, pb.gene_rate)["lnL"] for b in brs])
    out = torch.zeros(16, dtype=torch.float64, device="cuda")
    for i, b in enumerate(brs):
        eng.eval_device(b, out.data_ptr() + 8 * i, pb.gene_rate)
        if i == 4 + r % 3:
            eng.get_pmat(0, 0, 3)
        if i == 9:
            eng.eval(brs[2], pb.gene_rate)
        if big and i == 12 - r % 4:
            eng.flush()
    if big and r % 2:
        eng.flush()
        torch.cuda.current_stream().synchronize()
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    err = np.abs(got - want) / np.abs(want)
    if err.max() > 1e-13:
        bad += 1
        print("round", r, "mismatch at", np.nonzero(err > 1e-13)[0].tolist(), err.max(), flush=True)
    if big:
        print("round %d: %d states, %d taxa, %d patterns, %s%s" % (r, pb.n, pb.tree.n_tips, pb.n_patt, eng.kernel_name, ", one-rank communicator" if r % 3 == 2 else ""), flush=True)
    eng.close()
print("rounds", rounds, "bad", bad)
