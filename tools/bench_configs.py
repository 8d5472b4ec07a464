#!/usr/bin/env python3
"""Secondary measurements for the BASELINE.json configs that are parity-test cases rather than bench lines:
C2 (baseml GTR+G4, 32 taxa x 1e5 nucleotide patterns; contract bound: HBM, algorithmic 7 720 B/pattern),
C5 (HIV NSsites models: latency per evaluation incl. batched P(t)), and the NSsites sweep on the C4 data
(K = 1, 2, 3, 10, 11 omega classes over 1e6 codon patterns).  One JSON line per case."""
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "tests")]
from paml_amd import engine, synth  # noqa: E402


def timed(eng, branch, steps, warmup=3):
    for _ in range(warmup):
        eng.eval(branch)
    eng.profile(True)
    t0 = time.perf_counter()
    for _ in range(steps):
        r = eng.eval(branch)
    dt = (time.perf_counter() - t0) / steps
    p = eng.profile_read()
    eng.profile(False)
    return dt, {k: p[k] / max(1, p["n_evals"]) for k in ("ms_pmat", "ms_prune", "ms_reduce")}, r["lnL"]


def main():
    out = []
    # C2
    pb = synth.nuc_gtr_gamma_problem(n_tips=32, n_patt=100_000)
    eng = engine.engine_for(pb)
    dt, prof, lnl = timed(eng, pb.tree.branch, 50)
    bytes_pp = 8 * 4 * (30 + 29 + 1) * 4 + 32 + 8
    out.append(dict(case="C2 baseml GTR+G4 32x1e5", kernel=eng.kernel_name, ms_per_eval=dt * 1e3, patterns_per_s=pb.n_patt / dt,
                    prune_ms=prof["ms_prune"], algorithmic_GBps=bytes_pp * pb.n_patt / (prof["ms_prune"] * 1e-3) / 1e9,
                    hbm_peak_GBps=8000, lnL=lnl, **prof))
    # C2 at 1e7 patterns: past launch latency
    pb = synth.nuc_gtr_gamma_problem(n_tips=32, n_patt=4_000_000)
    eng = engine.engine_for(pb)
    dt, prof, lnl = timed(eng, pb.tree.branch, 10)
    out.append(dict(case="C2-shape 32x4e6", kernel=eng.kernel_name, ms_per_eval=dt * 1e3, patterns_per_s=pb.n_patt / dt,
                    prune_ms=prof["ms_prune"], algorithmic_GBps=bytes_pp * pb.n_patt / (prof["ms_prune"] * 1e-3) / 1e9,
                    hbm_peak_GBps=8000, lnL=lnl))
    # 20-state at scale (C3's kernel on a synthetic amino-acid-sized problem: 32 taxa x 1e5 patterns, 4 rate classes)
    import helpers
    pb = helpers.random_problem(20, 32, 100_000, K=4, seed=7)
    eng = engine.engine_for(pb)
    dt, prof, lnl = timed(eng, pb.tree.branch, 20)
    flops20 = (29 * 2 * 400 + 61 * 20 + 40) * 4.0 * pb.n_patt
    out.append(dict(case="20-state 32x1e5 K=4 (random reversible model)", kernel=eng.kernel_name, ms_per_eval=dt * 1e3,
                    patterns_per_s=pb.n_patt / dt, tflops=flops20 / (prof["ms_prune"] * 1e-3) / 1e12, fp64_valu_peak_tflops=78.6, lnL=lnl, **prof))
    # C5: HIV models, latency
    for name in ("hiv_m0", "hiv_m2a", "hiv_m8"):
        g = helpers.load_golden(name)
        pb = helpers.problem_from_golden(g)
        eng = engine.engine_for(pb)
        dt, prof, lnl = timed(eng, pb.tree.branch, 200)
        out.append(dict(case="C5 " + name, kernel=eng.kernel_name, K=pb.K, ms_per_eval=dt * 1e3, n_pmat=pb.K * 23, lnL=lnl,
                        golden_lnL=g["lnL"], **prof))
    # C5 gradient: np+1 = 26 branch-length sets in one launch (paml_amd_eval_batch) vs 26 single evaluations
    for name in ("hiv_m0", "hiv_m8"):
        g = helpers.load_golden(name)
        pb = helpers.problem_from_golden(g)
        eng = engine.engine_for(pb)
        nb = pb.tree.n_nodes
        br = np.tile(pb.tree.branch, (nb + 1, 1))
        for i in range(nb):
            br[i + 1, i] *= 1 + 1e-6
        for _ in range(3):
            eng.eval_batch(br)
        t0 = time.perf_counter()
        for _ in range(50):
            lb = eng.eval_batch(br)
        dtb = (time.perf_counter() - t0) / 50
        t0 = time.perf_counter()
        for _ in range(5):
            ls = [eng.eval(b)["lnL"] for b in br]
        dts = (time.perf_counter() - t0) / 5
        out.append(dict(case="C5 %s forward-difference gradient, %d evaluations" % (name, nb + 1), K=pb.K, kernel=eng.kernel_name,
                        ms_batched=dtb * 1e3, ms_one_by_one=dts * 1e3, us_per_eval_batched=dtb * 1e6 / (nb + 1),
                        max_abs_diff=float(np.max(np.abs(lb - np.array(ls))))))
    # NSsites sweep on C4 data
    base = synth.codon_m0_problem(n_tips=16, n_patt=1_000_000)
    for K in (1, 2, 3, 10, 11):
        pb = base if K == 1 else synth.codon_nssites_problem(base, 2.0, np.linspace(0.05, 1.5, K), np.full(K, 1.0 / K))
        eng = engine.engine_for(pb)
        dt, prof, lnl = timed(eng, pb.tree.branch, 5, warmup=2)
        flops = 98637.0 * K * pb.n_patt
        out.append(dict(case="C4 NSsites sweep K=%d" % K, kernel=eng.kernel_name, ms_per_eval=dt * 1e3,
                        pattern_classes_per_s=pb.n_patt * K / dt, tflops=flops / (prof["ms_prune"] * 1e-3) / 1e12, lnL=lnl, **prof))
        eng.close()
    # BEB (M2a) at scale: f(x_h|w) for the grid's 21 omegas (one 21-class evaluation) + the 10^4-point grid integral
    n1 = 10
    wgrid = np.concatenate([(np.arange(n1) + 0.5) / n1, [1.0], 1 + 10 * (np.arange(n1) + 0.5) / n1])
    pb = synth.codon_nssites_problem(base, 2.0, wgrid, np.full(21, 1.0 / 21))
    eng = engine.engine_for(pb)
    g = np.arange(n1 ** 4)
    ip0, ip1, ip2, ip3 = g // 1000, (g // 100) % 10, (g // 10) % 10, g % 10
    tri = ip0 * n1 + ip1
    ix = np.floor(np.sqrt(tri)).astype(int)
    iy = tri - ix * ix
    p0 = (1 + (iy // 2) * 3 + (iy % 2)) / (3.0 * n1)
    p1 = (1 + (n1 - 1 - ix) * 3 + (iy % 2)) / (3.0 * n1)
    pcl = np.stack([p0, p1, 1 - p0 - p1], axis=1)
    iw = np.stack([ip2, np.full_like(g, n1), n1 + 1 + ip3], axis=1).astype(np.int32)
    t0 = time.perf_counter()
    eng.eval(pb.tree.branch)
    t1 = time.perf_counter()
    r = eng.beb_grid(pcl, iw, wgrid)
    t2 = time.perf_counter()
    out.append(dict(case="BEB M2a on the C4 data (1e6 patterns): 21-class evaluation + 1e4-point grid integral", ms_eval=(t1 - t0) * 1e3,
                    ms_grid=(t2 - t1) * 1e3, grid_terms=float(n1 ** 4) * pb.n_patt * 3, terms_per_s=float(n1 ** 4) * pb.n_patt * 3 * 2 / (t2 - t1),
                    ln_fx=r["ln_fx"], sites_pr_gt_95=int((r["pr_last"] > 0.95).sum())))
    eng.close()
    for o in out:
        print(json.dumps(o))


if __name__ == "__main__":
    main()
