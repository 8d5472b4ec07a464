"""One-rank RCCL communicator through the engine ABI (probe: with and without torch in the process)."""
import faulthandler
import os
import sys
faulthandler.enable()
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "tests")]
if "--torch" in sys.argv:
    import torch
    torch.cuda.init()
    print("torch first", torch.cuda.device_count(), flush=True)
import helpers
from paml_amd import engine
pb = helpers.random_problem(61, 10, 3000, K=1, seed=1)
plain = engine.engine_for(pb).eval(pb.tree.branch)["lnL"]
print("plain", plain, flush=True)
uid = engine.comm_unique_id()
print("uid ok", len(uid), flush=True)
e = engine.engine_for(pb)
e.comm_init(0, 1, uid, pb.n_patt, 0)
print("comm ok", e.comm_info(), flush=True)
v = e.eval(pb.tree.branch)["lnL"]
print("eval", v, v == plain, flush=True)
e.comm_destroy()
print("done", flush=True)
