#!/usr/bin/env python3
"""Size sweep of the 61-state evaluation (codeml M0, synthetic data of bench.py's recipe): taxa x site patterns ->
ms per evaluation (back-to-back, lnL left on the device), algorithmic TFLOP/s of the pruning kernel, kernel chosen.
One JSON line per point."""
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO]
import torch  # noqa: E402
from paml_amd import engine, synth  # noqa: E402


def main():
    d_lnl = torch.zeros(1, dtype=torch.float64, device="cuda")
    for taxa in (8, 16, 32, 64):
        for n_patt in (1_000, 10_000, 100_000, 1_000_000, 4_000_000):
            if taxa * n_patt > 1.3e8:
                continue
            pb = synth.codon_m0_problem(n_tips=taxa, n_patt=n_patt)
            eng = engine.engine_for(pb)
            eng.set_stream(torch.cuda.current_stream().cuda_stream)
            br = pb.tree.branch
            for _ in range(3):
                eng.eval_device(br, d_lnl.data_ptr())
            torch.cuda.synchronize()
            steps = 50 if n_patt <= 100_000 else 10
            eng.profile(True)
            t0 = time.perf_counter()
            for _ in range(steps):
                eng.eval_device(br, d_lnl.data_ptr())
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps
            p = eng.profile_read()
            eng.profile(False)
            kms = p["ms_prune"] / max(1, p["n_evals"])
            flops = ((taxa - 3) * 2 * 61 * 61 + (2 * taxa - 3) * 61 + 122) * float(n_patt)
            print(json.dumps(dict(taxa=taxa, n_patt=n_patt, kernel=eng.kernel_name, ms_per_eval=round(dt * 1e3, 4), prune_ms=round(kms, 4),
                                  patterns_per_s=round(n_patt / dt), tflops=round(flops / (kms * 1e-3) / 1e12, 2))), flush=True)
            eng.close()


if __name__ == "__main__":
    main()
