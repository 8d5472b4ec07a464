"""Pattern sharding across ranks: site patterns are independent given the tree, the P(t) matrices, pi and the
class table, so each rank (one process per GPU) owns a contiguous block of patterns and ONE exchange step gives lnL.

On the GPUs the exchange lives INSIDE the engine (include/paml_amd.h: paml_amd_comm_init; RCCL over xGMI on a stream of the
engine's own, so that the next evaluation prunes while this one's partial sums are reduced): `sharded_engine` builds a rank's engine over its shard and joins the communicator, after which every eval* call
returns the total.  torch.distributed is only the courier of the 128-byte RCCL id (any backend).

The shard boundaries come from the C ABI (paml_amd_shard_bounds, host-only), and the helpers below restate the engine's
reduction scheme — one partial sum per chunk of patterns at the chunk's GLOBAL position, the ranks' zero-padded arrays added,
one fixed-order total — over any per-pattern evaluator, so that the same sharding is covered on CPU with the gloo backend
(tests/test_distributed_cpu.py) and the result is bit-identical for every world size."""
from __future__ import annotations

import numpy as np


def red_chunk(n_patt_global: int) -> int:
    """Patterns per partial sum: a function of the global pattern count alone (engine_state.h: red_chunk)."""
    return max(256, (((n_patt_global + 1023) // 1024) + 255) // 256 * 256)


def shard_bounds(n_patt: int, world: int, rank: int):
    """Contiguous [lo, hi) of `rank`, cut at multiples of the reduction chunk (paml_amd_shard_bounds)."""
    from . import engine
    first, count = engine.shard_bounds(n_patt, world, rank)
    return first, first + count


def _store_broadcast_bytes(payload, src=0):
    import torch.distributed as dist
    box = [payload]
    dist.broadcast_object_list(box, src=src)
    return box[0]


def sharded_engine(pb, flags=0, world=None, rank=None, force_comm=False):
    """This rank's engine over its pattern shard of `pb`, joined to the ranks' RCCL communicator.  world / rank default to the
    torch.distributed process group (which carries the RCCL id from rank 0 to the others); with one rank no communicator
    is made unless force_comm (a one-rank communicator: the collective path on a single GPU)."""
    from . import engine
    if world is None:
        import torch.distributed as dist
        world, rank = (dist.get_world_size(), dist.get_rank()) if dist.is_initialized() else (1, 0)
    lo, hi = shard_bounds(pb.n_patt, world, rank)      # (raises on EVERY rank alike when there are more ranks than reduction chunks)
    sub = pb.slice_patterns(lo, hi) if (lo, hi) != (0, pb.n_patt) else pb
    eng = engine.engine_for(sub, flags=flags | (engine.SHARD if sub is not pb else 0))
    uid = None
    if world > 1 or force_comm:
        uid = engine.comm_unique_id() if rank == 0 else None
        if world > 1:
            uid = _store_broadcast_bytes(uid, src=0)
    eng.comm_init(rank, world, uid, pb.n_patt, lo)
    return eng, (lo, hi)


# ---- the same scheme over a per-pattern evaluator (CPU / gloo coverage of the sharding and of the reduction order) ----------
def chunk_partials(lnf, weights, first_pattern, n_patt_global):
    """Zero-padded global array of per-chunk sums of w_h * lnf_h for the patterns [first_pattern, first_pattern + len(lnf))."""
    ch = red_chunk(n_patt_global)
    nb = -(-n_patt_global // ch)
    out = np.zeros(nb)
    assert first_pattern % ch == 0
    v = np.where(weights > 0, lnf, 0.0) * weights
    for c in range(-(-len(v) // ch)):
        out[first_pattern // ch + c] = float(np.sum(v[c * ch:(c + 1) * ch]))
    return out


def total_fixed_order(partials):
    """The fixed-order total of the (summed) partial array exactly as reduce_stage2 forms it on the device: 256 lanes sum the
    entries i, i + 256, ... in turn; each 64-lane wave then runs the xor butterfly (offsets 32 ... 1; every lane ends with the
    same bits, addition being commutative), and the four wave sums are combined as (w0 + w1) + (w2 + w3).  Pure additions, so
    the host reproduces the device's bits."""
    lanes = np.zeros(256)
    for i, v in enumerate(np.asarray(partials, dtype=np.float64)):
        lanes[i % 256] += v
    waves = []
    for w in range(4):
        v = lanes[64 * w:64 * w + 64].copy()
        idx = np.arange(64)
        for off in (32, 16, 8, 4, 2, 1):
            v = v + v[idx ^ off]
        waves.append(v[0])
    return float((waves[0] + waves[1]) + (waves[2] + waves[3]))


def sharded_lnl(pb, lnf_of_shard, world: int, rank: int):
    """lnL of `pb` from this rank's shard: lnf_of_shard(sub_problem) -> per-pattern log f_h; the exchange step is an all-reduce
    (sum) of the zero-padded chunk-partial array over torch.distributed (gloo on CPU)."""
    import torch
    import torch.distributed as dist
    lo, hi = shard_bounds(pb.n_patt, world, rank)
    part = np.zeros(-(-pb.n_patt // red_chunk(pb.n_patt)))
    if hi > lo:
        sub = pb.slice_patterns(lo, hi)
        part = chunk_partials(np.asarray(lnf_of_shard(sub)), sub.weights, lo, pb.n_patt)
    buf = torch.from_numpy(part)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dev = "cuda" if dist.get_backend() == "nccl" else "cpu"      # RCCL moves device memory only
        t = buf.to(dev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        buf = t.cpu()
    return total_fixed_order(buf.numpy()), (lo, hi)


def sharded_eval_branch(pb, eval_branch_shard, node_b, t, world: int, rank: int):
    """Branch-local lnL(t), dlnL/dt, d2lnL/dt2 (lfuntdd) over pattern shards: each rank's `eval_branch_shard(sub, node_b, t)
    -> (l, dl, ddl)` arrays are per-pattern sums, so the exchange step is one all-reduce of 3 * len(t) doubles
    ("count = 3 for eval_branch").  On the GPUs paml_amd_eval_branch does this itself."""
    import torch
    import torch.distributed as dist
    t = np.atleast_1d(np.asarray(t, dtype=np.float64))
    lo, hi = shard_bounds(pb.n_patt, world, rank)
    acc = np.zeros((3, len(t)))
    if hi > lo:
        acc[:] = np.stack(eval_branch_shard(pb.slice_patterns(lo, hi), node_b, t))
    buf = torch.from_numpy(acc)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
        tt = buf.to(dev)
        dist.all_reduce(tt, op=dist.ReduceOp.SUM)
        buf = tt.cpu()
    return buf.numpy()[0], buf.numpy()[1], buf.numpy()[2]
