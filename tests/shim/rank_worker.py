"""One rank of a world-size-N job with every rank on GPU 0 (test infrastructure; run by tests/test_multirank_gpu.py with
PAML_AMD_RCCL_LIB pointing at librccl_shim.so).  usage: rank_worker.py <rank> <world> <exchange dir> <case>
Builds the case's problem (seeded: the same on every rank), takes its pattern shard, joins the communicator — the id travels through
a file of the exchange directory — and writes what the engine's entry points return, as hexadecimal doubles, to out<rank>.json."""
import json
import os
import sys
import time

import torch  # noqa: F401  (before the engine library: torch ships its own copy of the HIP runtime, see tests/conftest.py)
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))


def problem(case):
    import helpers
    from paml_amd import synth
    if case == "codon_jit":      # the per-tree MFMA kernel (forced at this size), one class
        return synth.codon_m0_problem(n_tips=16, n_patt=6000), 2
    if case == "codon_big":      # large enough per rank for the runs of eval_device calls to alternate between two pruning streams
        return synth.codon_m0_problem(n_tips=16, n_patt=250_000), 0
    if case == "codon_k3":       # 61 states, three classes, interpreter kernels, scaling nodes
        return helpers.random_problem(61, 10, 3000, K=3, seed=11, scale_every=3), 0
    if case == "nuc_fused":      # 4 states: the fused kernel forms the partial sums itself
        return synth.nuc_gtr_gamma_problem(n_tips=32, n_patt=9000), 2
    if case == "aa20":
        return helpers.random_problem(20, 12, 5000, K=2, seed=12), 0
    raise SystemExit("unknown case " + case)


def run(pb, eng, n_dev=7):
    import torch
    out = {}
    br = pb.tree.branch
    out["eval"] = float(eng.eval(br, pb.gene_rate)["lnL"]).hex()
    d = torch.zeros(n_dev, dtype=torch.float64, device="cuda")
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    for i in range(n_dev):      # consecutive evaluations: the exchange step of i overlaps the pruning of i + 1 (two slots of partial sums)
        eng.eval_device(br * (1.0 + 0.01 * i), d.data_ptr() + 8 * i, pb.gene_rate)
    eng.flush()
    torch.cuda.current_stream().synchronize()      # (the stream alone, not the device: flush is what joins the collective stream)
    out["eval_device"] = [float(v).hex() for v in d.cpu().numpy()]
    B = np.stack([br, br * 1.1, br * 0.9])
    gr = None if pb.gene_rate is None else np.stack([pb.gene_rate] * 3)
    out["eval_batch"] = [float(v).hex() for v in eng.eval_batch(B, gene_rate=gr)]
    b = pb.tree.n_tips + 1
    ts = np.array([br[b], br[b] * 1.5 + 0.01])
    l, dl, ddl = eng.eval_branch(b, ts, br, pb.gene_rate)
    out["eval_branch"] = [float(v).hex() for v in np.concatenate([l, dl, ddl])]
    if pb.K > 1:      # what does not reduce to a sum of per-pattern terms: the auto-discrete-gamma chain over the sites, the BEB grid weights
        rng = np.random.default_rng(5)
        MK = rng.dirichlet(np.ones(pb.K), size=pb.K)
        pose = rng.integers(0, pb.n_patt, size=4000).astype(np.int32)      # GLOBAL pattern indices, the same on every rank
        out["eval_adg"] = float(eng.eval_adg(br, MK, pose, pb.gene_rate)).hex()
        eng.eval(br, pb.gene_rate, want_fhk=True)
        ncls, ngrid = 3, 60
        pcl = rng.dirichlet(np.ones(ncls), size=ngrid)
        iw = rng.integers(0, pb.K, size=(ngrid, ncls)).astype(np.int32)
        b = eng.beb_grid(pcl, iw, np.linspace(0.2, 3.0, pb.K))
        c = eng.beb_grid_classes(pcl, iw)
        out["beb"] = dict(ln_fx=b["ln_fx"], pr_last=b["pr_last"].tolist(), mean_w=b["mean_w"].tolist(), sd_w=b["sd_w"].tolist(),
                          ln_fx_classes=c["ln_fx"], post=c["post"].tolist())
    out["eval_again"] = float(eng.eval(br, pb.gene_rate)["lnL"]).hex()
    out["kernel"] = eng.kernel_name
    return out


def main():
    rank, world, xdir, case = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    from paml_amd import distributed, engine
    real = os.environ.get("PAML_AMD_WORKER_REAL_RCCL") == "1"      # one GPU per rank, the real collective library (multi-GPU boxes)
    if real and world > 1:
        torch.cuda.set_device(rank)
        assert engine.lib().paml_amd_set_device(rank) == 0
    pb, flags = problem(case)
    if world == 1:      # the reference run: one engine over everything, no communicator
        eng = engine.engine_for(pb, flags=flags)
        res = run(pb, eng)
    else:
        idfile = os.path.join(xdir, "id")
        if rank == 0:
            uid = engine.comm_unique_id()
            with open(idfile + ".tmp", "wb") as f:
                f.write(uid)
            os.rename(idfile + ".tmp", idfile)
        else:
            t0 = time.time()
            while not os.path.exists(idfile):
                time.sleep(0.05)
                if time.time() - t0 > 120:
                    raise SystemExit("rank %d: no id from rank 0" % rank)
            uid = open(idfile, "rb").read()
        lo, hi = distributed.shard_bounds(pb.n_patt, world, rank)
        eng = engine.engine_for(pb.slice_patterns(lo, hi), flags=flags | engine.SHARD)      # (a shard: kernels not chosen by its own size)
        eng.comm_init(rank, world, uid, pb.n_patt, lo)
        res = run(pb, eng)
        res["shard"] = [lo, hi]
        if real:      # what the exchange step did on real hardware: timed events over a run of evaluations, and which library it was
            eng.comm_stats(True)
            d = torch.zeros(16, dtype=torch.float64, device="cuda")
            for i in range(16):
                eng.eval_device(pb.tree.branch, d.data_ptr() + 8 * i, pb.gene_rate)
            eng.flush()
            torch.cuda.current_stream().synchronize()
            res["comm_stats"] = eng.comm_stats(False, read=True)
            res["comm_library"] = engine.comm_library()
            res["device"] = torch.cuda.current_device()
            res["run16"] = [float(v).hex() for v in d.cpu().numpy()]
    eng.close()
    with open(os.path.join(xdir, "out%d.json" % rank), "w") as f:
        json.dump(res, f)


if __name__ == "__main__":
    main()
