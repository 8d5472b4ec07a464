#!/usr/bin/env python3
"""Small data sets (BASELINE configs[0], [2], [4]): what one evaluation consists of.  Runs N evaluations of a golden case back to back and
N with the scalar read back, prints the wall time of each loop; under `rocprofv3 --kernel-trace` tools/small_timeline_digest.py turns
the trace into the kernels of an evaluation with their durations and the gaps between them.
usage: python tools/small_timeline.py [hiv_m0|hiv_m8|stewart|brown] [n]"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "tests")]
import torch  # noqa: E402
import helpers  # noqa: E402
from paml_amd import engine  # noqa: E402

CASES = {"hiv_m0": "hiv_m0", "hiv_m8": "hiv_m8", "stewart": "stewart_lg_g4", "brown": "brown_hky85"}
name = sys.argv[1] if len(sys.argv) > 1 else "hiv_m0"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
g = helpers.load_golden(CASES[name])
pb = helpers.problem_from_golden(g)
eng = engine.engine_for(pb)
br = pb.tree.branch
for _ in range(10):
    v = eng.eval(br, pb.gene_rate)["lnL"]
torch.cuda.synchronize()
d = torch.zeros(n, dtype=torch.float64, device="cuda")
eng.set_stream(torch.cuda.current_stream().cuda_stream)
t0 = time.perf_counter()
for i in range(n):
    eng.eval_device(br, d.data_ptr() + 8 * i, pb.gene_rate)
eng.flush(); torch.cuda.synchronize()
t1 = time.perf_counter()
for i in range(n):
    v = eng.eval(br, pb.gene_rate)["lnL"]
torch.cuda.synchronize()
t2 = time.perf_counter()
print("%s kernel %s lnL %.6f (golden %.6f): back to back %.1f us per evaluation, with the read-back %.1f us" %
      (name, eng.kernel_name, v, g["lnL"], (t1 - t0) / n * 1e6, (t2 - t1) / n * 1e6))
