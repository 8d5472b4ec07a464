"""N>1 path on CPU: world_size-2 gloo runs of the pattern-sharding and reduction scheme the engine uses across GPUs
(paml_amd_shard_bounds from the C ABI; per-chunk partial sums at global positions, the ranks' zero-padded arrays added, one
fixed-order total).  The per-shard evaluator here is the oracle; on the GPU box it is the engine (tests/test_engine_gpu.py
runs the RCCL path in a one-rank communicator)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

import helpers
import oracle
from paml_amd import distributed, engine, synth


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


N_PATT = 3000      # chunk 256 -> 12 chunks, 6 per rank


def _problem(genes=1):
    if genes > 1:      # option G: three genes with their own frequencies, rates and eigen systems
        pb = helpers.random_problem(4, 9, N_PATT, K=2, seed=17, n_genes=genes)
        pb.gene_off = np.array([0, 700, 1500, N_PATT], dtype=np.int32)      # the first two genes end inside rank 0's block [0, 1536)
        return pb
    return synth.nuc_gtr_gamma_problem(n_tips=12, n_patt=N_PATT, seed=5)


def _worker(rank, world, port, out, genes=1):
    import torch.distributed as dist
    for p in (helpers.REPO, os.path.join(helpers.REPO, "oracle"), os.path.join(helpers.REPO, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import oracle as orc
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    pb = _problem(genes)
    lnl, (lo, hi) = distributed.sharded_lnl(pb, lambda sub: orc.evaluate(sub, want_lnf=True)["lnf"], world, rank)
    if genes > 1:
        out[rank] = (lnl, lo, hi)
    else:
        b = pb.tree.n_tips + 2
        tt = np.array([pb.tree.branch[b], 0.2])
        l, dl, ddl = distributed.sharded_eval_branch(pb, lambda sub, nb, ts: orc.eval_branch(sub, nb, ts), b, tt, world, rank)
        out[rank] = (lnl, lo, hi, l.tolist(), dl.tolist(), ddl.tolist())
    dist.barrier()
    dist.destroy_process_group()


def test_shard_bounds_cover_and_align():
    for n, w in [(1000, 2), (1_000_000, 8), (1_000_000, 3), (79, 1), (600, 3), (4_000_000, 8), (100_000, 8)]:
        ch = distributed.red_chunk(n)
        assert ch % 256 == 0 and -(-n // ch) <= 1024
        spans = [distributed.shard_bounds(n, w, r) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        for (a, b), (c, d) in zip(spans, spans[1:]):
            assert b == c and a <= b
        assert all(lo % ch == 0 and hi > lo for lo, hi in spans)      # no rank without patterns
        sizes = [hi - lo for lo, hi in spans]
        assert max(sizes) - min(sizes) <= ch or n < ch * w      # as even as whole chunks allow
    # the C ABI rejects nonsense
    assert engine.lib().paml_amd_shard_bounds(0, 1, 0, None, None) != 0
    # more ranks than reduction chunks would leave ranks without patterns (and the others waiting for them in a collective call):
    # refused for EVERY rank alike, so that all ranks of a job fail together
    import ctypes as C
    for n, w in [(79, 4), (129, 2), (200, 2), (1000, 8)]:
        assert engine.max_ranks(n) < w
        for r in range(w):
            first, count = C.c_long(), C.c_long()
            assert engine.lib().paml_amd_shard_bounds(n, w, r, C.byref(first), C.byref(count)) != 0
        with pytest.raises(engine.EngineError):
            distributed.shard_bounds(n, w, 0)
    assert engine.max_ranks(1_000_000) >= 8 and engine.max_ranks(100_000) >= 8 and engine.max_ranks(79) == 1


def test_rccl_stand_in_of_the_tests_builds_and_exports_what_the_engine_binds():
    """tests/shim/librccl_shim.so (the shared-memory stand-in that lets the -m gpu tier run world > 1 on one GPU) builds with
    hipcc and exports exactly the five symbols engine_state.h's Rccl binds; the id round trip needs no GPU."""
    import ctypes as C
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    subprocess.check_call(["make", "-C", os.path.join(here, "shim")], stdout=subprocess.DEVNULL)
    L = C.CDLL(os.path.join(here, "shim", "librccl_shim.so"))
    for sym in ("ncclGetUniqueId", "ncclCommInitRank", "ncclCommDestroy", "ncclAllReduce", "ncclGetErrorString"):
        assert hasattr(L, sym)
    src = open(os.path.join(here, "..", "paml_amd", "csrc", "engine_state.h")).read()
    assert src.count('dlsym(h, "nccl') == 5
    uid = C.create_string_buffer(128)
    assert L.ncclGetUniqueId(uid) == 0 and uid.raw.startswith(b"paml_amd_rccl_shim:/")
    os.unlink("/dev/shm" + uid.raw.split(b":")[1].rstrip(b"\0").decode())


def test_reduction_scheme_is_independent_of_world_size():
    """Summing the ranks' zero-padded chunk-partial arrays and taking one fixed-order total gives the SAME bits for 1, 2, 3, 5 ranks."""
    pb = _problem()
    lnf = oracle.evaluate(pb, want_lnf=True)["lnf"]
    ref = distributed.total_fixed_order(distributed.chunk_partials(lnf, pb.weights, 0, pb.n_patt))
    for world in (2, 3, 5):
        tot = np.zeros(-(-pb.n_patt // distributed.red_chunk(pb.n_patt)))
        for r in range(world):
            lo, hi = distributed.shard_bounds(pb.n_patt, world, r)
            if hi > lo:
                tot += distributed.chunk_partials(lnf[lo:hi], pb.weights[lo:hi], lo, pb.n_patt)
        assert distributed.total_fixed_order(tot) == ref      # bit-identical
    assert abs(ref - float(np.dot(lnf, pb.weights))) <= 1e-12 * abs(ref)


def test_two_rank_gloo_matches_single():
    pb = _problem()
    res = oracle.evaluate(pb, want_lnf=True)
    single = distributed.total_fixed_order(distributed.chunk_partials(res["lnf"], pb.weights, 0, pb.n_patt))
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    assert set(out.keys()) == {0, 1}
    for r in (0, 1):
        assert out[r][0] == single                                     # bit-identical to the one-rank total
        assert abs(out[r][0] - res["lnL"]) <= 1e-12 * abs(res["lnL"])
    assert out[0][2] == out[1][1] and out[0][2] % distributed.red_chunk(pb.n_patt) == 0      # contiguous, chunk-aligned
    # branch-local evaluation: the 3 x n_t sums all-reduce to the single-process values
    b = pb.tree.n_tips + 2
    rl, rdl, rddl = oracle.eval_branch(pb, b, np.array([pb.tree.branch[b], 0.2]))
    for r in (0, 1):
        assert np.allclose(out[r][3], rl, rtol=1e-12) and np.allclose(out[r][4], rdl, rtol=1e-9, atol=1e-9)
        assert np.allclose(out[r][5], rddl, rtol=1e-9, atol=1e-8)
    assert abs(rl[0] - res["lnL"]) <= 1e-11 * abs(res["lnL"])


def test_two_rank_gloo_with_several_genes():
    """Option G data shard the same way (gene boundaries stay where they are in the global pattern range): a rank holds the
    part of every gene inside its block — here rank 1 holds nothing of the first gene — and the total is the one-rank total bit for bit."""
    pb = _problem(3)
    assert pb.n_genes == 3
    res = oracle.evaluate(pb, want_lnf=True)
    single = distributed.total_fixed_order(distributed.chunk_partials(res["lnf"], pb.weights, 0, pb.n_patt))
    lo1, hi1 = distributed.shard_bounds(pb.n_patt, 2, 1)
    sub = pb.slice_patterns(lo1, hi1)
    assert sub.gene_off[0] == 0 and sub.gene_off[-1] == hi1 - lo1 and sub.gene_off[1] == 0 and (np.diff(sub.gene_off) >= 0).all()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), out, 3), nprocs=2, join=True)
    for r in (0, 1):
        assert out[r][0] == single
        assert abs(out[r][0] - res["lnL"]) <= 1e-12 * abs(res["lnL"])
