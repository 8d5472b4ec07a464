#!/usr/bin/env python3
"""4-state kernel probe: C2 (32 taxa x 1e5 nucleotide patterns, GTR+G4) and the same shape at 4e6 patterns — per-evaluation time
in the three call styles (eval with host read-back, eval_device back to back) and the kernel's own time (HIP events), checked
against the oracle on a slice.  PAML_AMD_NO_FUSED=1 selects the round-1 kernel (classes outermost, tips gathered from L2)."""
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "tests"), os.path.join(REPO, "oracle")]
import torch  # noqa: E402,F401  (its HIP runtime first, see tests/conftest.py)
from paml_amd import engine, synth  # noqa: E402


def run(n_patt, steps):
    pb = synth.nuc_gtr_gamma_problem(n_tips=32, n_patt=n_patt)
    eng = engine.engine_for(pb)
    br = pb.tree.branch
    for _ in range(5):
        r = eng.eval(br)
    t0 = time.perf_counter()
    for _ in range(steps):
        r = eng.eval(br)
    dt_sync = (time.perf_counter() - t0) / steps
    d = torch.zeros(steps, dtype=torch.float64, device="cuda")
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    for i in range(5):
        eng.eval_device(br, d.data_ptr())
    eng.flush(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        eng.eval_device(br, d.data_ptr() + 8 * i)
    eng.flush(); torch.cuda.synchronize()
    dt_dev = (time.perf_counter() - t0) / steps
    assert float(d[-1].item()) == r["lnL"]
    eng.profile(True)
    for _ in range(20):
        eng.eval(br)
    p = eng.profile_read()
    eng.profile(False)
    k = {q: p[q] / p["n_evals"] for q in ("ms_pmat", "ms_prune", "ms_reduce")}
    flops = (29 * 32 + 61 * 4 + 8) * 4.0 * n_patt
    out = dict(case="32 taxa x %d patterns GTR+G4" % n_patt, kernel=eng.kernel_name, ms_eval_sync=dt_sync * 1e3, ms_eval_device=dt_dev * 1e3,
               lnL=r["lnL"], valu_tflops=flops / (k["ms_prune"] * 1e-3) / 1e12, valu_frac=flops / (k["ms_prune"] * 1e-3) / 1e12 / 78.6, **k)
    if n_patt <= 200_000 and not os.environ.get("C2_NOCHECK"):
        import oracle
        ref = oracle.evaluate(pb)
        out["oracle_rel_diff"] = abs(r["lnL"] - ref["lnL"]) / abs(ref["lnL"])
        o2 = eng.eval(br, want_lnf=True, want_fhk=True)
        out["max_lnf_diff"] = float(np.max(np.abs(o2["lnf"] - ref["lnf"])))
    eng.close()
    return out


if __name__ == "__main__":
    # an argument: that pattern count alone (counter / kernel-statistics runs: one workload per run, so that a per-kernel average IS that workload's)
    sizes = ((int(sys.argv[1]), 200),) if len(sys.argv) > 1 else ((100_000, 200), (4_000_000, 20))
    for n, st in sizes:
        print(json.dumps(run(n, st)), flush=True)
