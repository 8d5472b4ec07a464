"""Pattern sharding across ranks: site patterns are independent given the tree, the P(t) matrices,
pi and the class table, so each rank owns a contiguous block of patterns, evaluates its partial
sum_h w_h log f_h, and ONE exchange step — an all-reduce of that f64 scalar — gives lnL.
torch.distributed backend "nccl" is RCCL over xGMI on the MI355X node; "gloo" covers the same code on CPU."""
from __future__ import annotations

import numpy as np


def shard_bounds(n_patt: int, world: int, rank: int, align: int = 128):
    """Contiguous [lo, hi) for `rank`; shard starts are multiples of `align` (the kernels' tile) so no tile straddles ranks."""
    per = -(-n_patt // world)
    per = -(-per // align) * align
    lo = min(n_patt, rank * per)
    hi = min(n_patt, lo + per)
    return lo, hi


def allreduce_lnl(local_lnl, device=None):
    """Sum the per-rank partial lnL.  `local_lnl` may be a python float or a 1-element torch tensor (on the GPU for RCCL)."""
    import torch
    import torch.distributed as dist
    t = local_lnl if isinstance(local_lnl, torch.Tensor) else torch.tensor([float(local_lnl)], dtype=torch.float64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def sharded_eval(pb, evaluate_shard, world: int, rank: int):
    """Evaluate this rank's shard with `evaluate_shard(sub_problem) -> float` and all-reduce.  Empty shards contribute 0."""
    lo, hi = shard_bounds(pb.n_patt, world, rank)
    local = 0.0
    if hi > lo:
        local = float(evaluate_shard(pb.slice_patterns(lo, hi)))
    return float(allreduce_lnl(local).item()), (lo, hi)


def sharded_eval_branch(pb, eval_branch_shard, node_b, t, world: int, rank: int):
    """Branch-local lnL(t), dlnL/dt, d2lnL/dt2 (lfuntdd) over pattern shards: each rank's `eval_branch_shard(sub, node_b, t)
    -> (l, dl, ddl)` arrays are per-pattern sums, so the exchange step is one all-reduce of 3 * len(t) doubles
    ("count = 3 for eval_branch")."""
    import torch
    import torch.distributed as dist
    t = np.atleast_1d(np.asarray(t, dtype=np.float64))
    lo, hi = shard_bounds(pb.n_patt, world, rank)
    acc = np.zeros((3, len(t)))
    if hi > lo:
        acc[:] = np.stack(eval_branch_shard(pb.slice_patterns(lo, hi), node_b, t))
    buf = torch.from_numpy(acc)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    return buf.numpy()[0], buf.numpy()[1], buf.numpy()[2]
