"""N>1 path on CPU: world_size-2 gloo run of the pattern-sharding + scalar all-reduce logic used by bench.py and the
multi-GPU integration (the per-shard evaluator here is the oracle; on the GPU box it is the engine)."""
import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

import helpers
import oracle
from paml_amd import distributed, synth


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_patt, out):
    import torch.distributed as dist
    for p in (helpers.REPO, os.path.join(helpers.REPO, "oracle"), os.path.join(helpers.REPO, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import oracle as orc
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    pb = synth.nuc_gtr_gamma_problem(n_tips=12, n_patt=n_patt, seed=5)
    lnl, (lo, hi) = distributed.sharded_eval(pb, lambda sub: orc.evaluate(sub, want_lnf=False)["lnL"], world, rank)
    b = pb.tree.n_tips + 2
    tt = np.array([pb.tree.branch[b], 0.2])
    l, dl, ddl = distributed.sharded_eval_branch(pb, lambda sub, nb, ts: orc.eval_branch(sub, nb, ts), b, tt, world, rank)
    out[rank] = (lnl, lo, hi, l.tolist(), dl.tolist(), ddl.tolist())
    dist.barrier()
    dist.destroy_process_group()


def test_shard_bounds_cover_and_align():
    for n, w in [(1000, 2), (1_000_000, 8), (79, 4), (129, 2)]:
        spans = [distributed.shard_bounds(n, w, r) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        for (a, b), (c, d) in zip(spans, spans[1:]):
            assert b == c and a <= b
        assert all(lo % 128 == 0 for lo, hi in spans if hi > lo)


def test_two_rank_gloo_matches_single():
    n_patt = 1000
    pb = synth.nuc_gtr_gamma_problem(n_tips=12, n_patt=n_patt, seed=5)
    ref = oracle.evaluate(pb, want_lnf=False)["lnL"]
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), n_patt, out), nprocs=2, join=True)
    assert set(out.keys()) == {0, 1}
    for r in (0, 1):
        assert abs(out[r][0] - ref) <= 1e-12 * abs(ref)
    assert out[0][2] == out[1][1]           # shards are contiguous
    # branch-local evaluation: the 3 x n_t sums all-reduce to the single-process values
    b = pb.tree.n_tips + 2
    rl, rdl, rddl = oracle.eval_branch(pb, b, np.array([pb.tree.branch[b], 0.2]))
    for r in (0, 1):
        assert np.allclose(out[r][3], rl, rtol=1e-12) and np.allclose(out[r][4], rdl, rtol=1e-9, atol=1e-9)
        assert np.allclose(out[r][5], rddl, rtol=1e-9, atol=1e-8)
    assert abs(rl[0] - ref) <= 1e-11 * abs(ref)
