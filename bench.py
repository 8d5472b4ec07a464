#!/usr/bin/env python3
"""bench.py — site-patterns/sec per lnL evaluation, codeml M0 (61 states), on N MI355X.

A "step" is one com.plfun-equivalent evaluation over the rank's resident pattern shard: batched P(t)
for all 29 branches, fused FP64-MFMA pruning over every pattern, root/log/weighted-sum reduction,
(N>1: RCCL all-reduce of the scalar lnL over xGMI), and the scalar read back to the host — i.e. what
the optimiser in the reference waits for on every function call (codeml.c:748).  Inputs (tip codes,
weights, tree program) are resident in HBM before the timed region; only branch lengths (232 B) and
the lnL cross PCIe per step.

Workload (BASELINE.json configs[3]): 16 taxa x 10^6 synthetic codon patterns PER GPU (weak scaling:
patterns are independent, each rank owns a contiguous shard), M0 kappa=2 omega=0.4, F3x4 pi, fixed
parameters, seeded generator paml_amd.synth.  launched as
    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

# FP64 MFMA dense peak of MI355X: 256 CU x 4 SIMD x 32 FLOP/clk (v_mfma_f64_16x16x4_f64 = 2048 FLOP
# per 64 clk) x 2.4 GHz = 78.6 TFLOP/s (AMD datasheet "FP64 matrix 78.6 TF"); MI355X_MICROARCH.md lists
# no FP64 row, tools/mfma_f64_peak.hip measures the ceiling on the box.
FP64_MFMA_PEAK_TFLOPS = 78.6
# HBM bytes one launch of the pruning kernel moves at the default workload (16 taxa x 1e6 patterns), from the PMC
# counters as MI355X_MICROARCH.md prescribes: 2 x FETCH_SIZE (gfx950 wide-read correction, upper bound) + WRITE_SIZE,
# separate rocprofv3 --pmc passes; numbers and command in profiles/r01_pmc_summary.txt.
HBM_TRAFFIC_BYTES_PER_LAUNCH = int((2 * 20058.4 + 7812.5) * 1024)


def algorithmic_flops_per_pattern(n, n_tips):
    """Bi*2n^2 + B*n + 2n with Bi = ns-3 internal-son branches, B = 2ns-3 branches."""
    return (n_tips - 3) * 2 * n * n + (2 * n_tips - 3) * n + 2 * n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--patterns", type=int, default=1_000_000, help="site patterns per GPU")
    ap.add_argument("--taxa", type=int, default=16)
    ap.add_argument("--classes", type=int, default=1, help=">1: NSsites-style omega classes (M0 when 1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=100_000)
    ap.add_argument("--allreduce-bucket", type=int, default=8,
                    help="N > 1: lnL values of this many consecutive evaluations share one all-reduce (1 = one collective per evaluation)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE=%d but --gpus %d: launch with torch.distributed.run --nproc-per-node %d"
                         % (world, args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the engine has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    # (PAML_AMD_BENCH_FORCE_DIST=1 runs the collective path in a 1-rank group: a check of that code on a single-GPU box)
    use_dist = world > 1 or os.environ.get("PAML_AMD_BENCH_FORCE_DIST") == "1"
    if use_dist:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29511")
            os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from paml_amd import engine, synth
    if not os.path.exists(engine.LIB_PATH):
        engine.build()

    # this rank's shard: its own seeded block of the synthetic alignment
    pb = synth.codon_m0_problem(n_tips=args.taxa, n_patt=args.patterns, seed=20260926 + rank)
    if args.classes > 1:
        K = args.classes
        omegas = np.linspace(0.05, 1.5, K)
        pb = synth.codon_nssites_problem(pb, 2.0, omegas, np.full(K, 1.0 / K))
    eng = engine.engine_for(pb)
    stream = torch.cuda.current_stream()
    eng.set_stream(stream.cuda_stream)
    d_lnl = torch.zeros(args.warmup + args.steps, dtype=torch.float64, device="cuda")      # one slot per evaluation
    branch = pb.tree.branch.copy()
    pending = []
    bucket = max(1, args.allreduce_bucket)
    unsent = [0]                                   # first slot not yet handed to a collective

    def flush(upto):
        # the exchange step: sum the ranks' partial lnL of the evaluations [unsent, upto) — a bucket of 8-byte scalars, as a
        # gradient's independent evaluations need their totals only together
        if use_dist and upto > unsent[0]:
            if os.environ.get("PAML_AMD_BENCH_SYNC_ALLREDUCE") == "1":      # A/B switch: the collective in line with the compute stream
                dist.all_reduce(d_lnl[unsent[0]:upto])
            else:
                pending.append(dist.all_reduce(d_lnl[unsent[0]:upto], async_op=True))
        unsent[0] = upto

    def step(i):
        # one likelihood evaluation of the whole alignment: P(t) for every branch, the pruning kernel, the weighted
        # reduction, and (N > 1) the all-reduce of the scalar.  Everything is enqueued on the stream; the lnL value stays
        # on the device, so consecutive evaluations run back to back (a gradient's evaluations are independent of each
        # other's results) and the host only synchronises at the fences around the timed region.  Each evaluation has its
        # own result slot; the slots of `bucket` consecutive evaluations go through one asynchronous all-reduce (RCCL's
        # stream, ordered after the last of them by an event), so neither the collective's latency nor its kernel sits
        # between two evaluations.
        eng.eval_device(branch, d_lnl.data_ptr() + 8 * i)
        if i + 1 - unsent[0] >= bucket:
            flush(i + 1)

    def fence():
        while pending:
            pending.pop().wait()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    flush(args.warmup)
    fence()
    lnl_warm = float(d_lnl[args.warmup - 1].item()) if args.warmup else None

    eng.profile(True)
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    flush(args.warmup + args.steps)
    fence()
    dt = time.perf_counter() - t0
    lnl = float(d_lnl[-1].item())
    if lnl_warm is None:
        lnl_warm = lnl
    if not bool(torch.all(torch.abs(d_lnl - lnl) <= 1e-12 * abs(lnl))):
        raise SystemExit("bench: lnL differs between evaluations: %r" % (d_lnl.tolist(),))
    if not abs(lnl - lnl_warm) <= 1e-12 * abs(lnl_warm):      # same inputs every step (the sum order of an all-reduce may differ)
        raise SystemExit("bench: lnL changed between evaluations (%r vs %r)" % (lnl, lnl_warm))
    prof = eng.profile_read()
    eng.profile(False)
    if use_dist:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    if rank == 0:
        total_patterns = args.patterns * world
        value = total_patterns * args.steps / dt
        flops_pp = algorithmic_flops_per_pattern(pb.n, args.taxa) * pb.K
        default_workload = args.taxa == 16 and args.patterns == 1000000
        ms_kernel = prof["ms_prune"] / max(1, prof["n_evals"])
        achieved = flops_pp * args.patterns / (ms_kernel * 1e-3) / 1e12
        out = {
            "metric": "site-patterns/sec per lnL eval (codeml M0, 61 states)",
            "value": value, "unit": "site-patterns/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "codeml M0 61-state, %d taxa x %d synthetic codon patterns per GPU (BASELINE configs[3])"
                                   % (args.taxa, args.patterns),
                       "classes": pb.K, "kernel": eng.kernel_name, "parallelism": "pattern-shard x%d" % world},
            "lnL": lnl,
            "roofline": {"bound": "mfma", "kernel": "prune_mfma64", "achieved": achieved, "peak": FP64_MFMA_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": achieved / FP64_MFMA_PEAK_TFLOPS,
                         "traffic": HBM_TRAFFIC_BYTES_PER_LAUNCH if default_workload else None,
                         "traffic_note": "HBM bytes per launch, rocprofv3 --pmc FETCH_SIZE (x2 gfx950 correction) + WRITE_SIZE, "
                                         "separate passes: profiles/r01_pmc_summary.txt",
                         "flop_per_pattern": flops_pp, "kernel_ms": ms_kernel,
                         # (consecutive evaluations build P(t) on a side stream under the previous kernel's last round: the
                         #  event pair around it then spans its wait for free CUs, which is not kernel time)
                         "pmat_ms": (prof["ms_pmat"] / max(1, prof["n_evals"])) if prof["ms_pmat"] < 0.5 * prof["ms_prune"] else None,
                         "reduce_ms": prof["ms_reduce"] / max(1, prof["n_evals"])},
        }
        if not args.no_cpu_baseline and world == 1:      # reported baseline: rank 0 at N=1 only
            out["cpu_baseline"] = cpu_baseline(pb, args.cpu_sample)
            out["speedup_vs_cpu_1core"] = value / world / out["cpu_baseline"]["value"]
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def usable_cores():
    """Host cores this process may really use: the affinity mask, capped by the cgroup CPU quota (cpu.max) if there is one."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period))))
    except (OSError, ValueError):
        pass
    return n


def reference_binary_rate(pb):
    """The unmodified reference program (oracle/_ref/codeml, built by oracle/Makefile from the reference's own sources) on the
    first patterns of the same workload: one likelihood evaluation at fixed parameters (fix_blength = 2, kappa and omega fixed),
    wall time of two sample sizes, their difference isolating the per-pattern cost from start-up and file I/O.
    None when the binary is not there (it is git-ignored and travels only as a built file)."""
    import shutil
    import subprocess
    import tempfile
    exe = os.path.join(REPO, "oracle", "_ref", "codeml")
    if not (os.path.isfile(exe) and os.access(exe, os.X_OK)) or pb.n != 61 or pb.K != 1:
        return None
    from paml_amd import synth
    times = {}
    try:
        for n in (5000, 40000):
            n = min(n, pb.n_patt)
            sub = pb.slice_patterns(0, n)
            d = tempfile.mkdtemp(prefix="paml_amd_ref_")
            try:
                synth.write_pattern_file(os.path.join(d, "seq.txt"), sub.z, sub.weights, "codon")
                with open(os.path.join(d, "tree.txt"), "w") as f:
                    f.write(" %d 1\n%s\n" % (pb.tree.n_tips, pb.tree.newick()))
                with open(os.path.join(d, "codeml.ctl"), "w") as f:
                    f.write("seqfile = seq.txt\ntreefile = tree.txt\noutfile = mlc\nnoisy = 0\nverbose = 0\nrunmode = 0\nseqtype = 1\n"
                            "CodonFreq = 2\nmodel = 0\nNSsites = 0\nicode = 0\nfix_kappa = 1\nkappa = 2\nfix_omega = 1\nomega = 0.4\n"
                            "fix_alpha = 1\nalpha = 0\ngetSE = 0\nRateAncestor = 0\ncleandata = 1\nfix_blength = 2\nmethod = 0\n")
                t0 = time.perf_counter()
                r = subprocess.run([exe, "codeml.ctl"], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, input=b"\n" * 20, timeout=300)
                times[n] = time.perf_counter() - t0
                if b"lnL" not in r.stdout and not os.path.exists(os.path.join(d, "mlc")):
                    return None
            finally:
                shutil.rmtree(d, ignore_errors=True)
    except Exception:
        return None
    (n0, t0), (n1, t1) = sorted(times.items())
    if n1 <= n0 or t1 <= t0:
        return None
    return {"value": (n1 - n0) / (t1 - t0), "unit": "site-patterns/s", "cores": 1,
            "sample": "oracle/_ref/codeml (the unmodified reference, gcc -O3), one lnL evaluation at fixed parameters: wall time of %d "
                      "patterns (%.2f s) minus that of %d (%.2f s)" % (n1, t1, n0, t0)}


def cpu_baseline(pb, sample):
    """The oracle's single-thread restatement of the reference loop nest (kind "port"), timed on this
    box's host cores over the first `sample` patterns of the same workload."""
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import oracle
    sample = min(sample, pb.n_patt)
    sub = pb.slice_patterns(0, sample)
    oracle.evaluate(sub.slice_patterns(0, min(2000, sample)), want_lnf=False)   # warm-up / page-in
    reps = 0
    t0 = time.perf_counter()
    while True:
        oracle.evaluate(sub, want_lnf=False)
        reps += 1
        el = time.perf_counter() - t0
        if el > 10.0 or reps >= 8:
            break
    one = {"value": sample * reps / el, "unit": "site-patterns/s", "cores": 1, "kind": "port",
           "sample": "%d evals over the first %d patterns of the workload, oracle/cpu_ref.c, gcc -O3, 1 thread (%d host cores present)"
                     % (reps, sample, os.cpu_count() or 0)}
    # the same code with the patterns cut into blocks spread over every host core, each thread walking the whole tree
    # for its block; the whole workload, a few evaluations, bounded to ~10 s
    ncores = usable_cores()
    oracle.evaluate_blocked(pb.slice_patterns(0, min(pb.n_patt, 4096 * ncores)), ncores)      # thread start-up
    reps = 0
    t0 = time.perf_counter()
    while True:
        oracle.evaluate_blocked(pb, ncores)
        reps += 1
        el = time.perf_counter() - t0
        if el > 8.0 or reps >= 5:
            break
    one["all_cores"] = {"value": pb.n_patt * reps / el, "unit": "site-patterns/s", "cores": ncores,
                        "sample": "%d evals over all %d patterns, blocks of 512 patterns over %d OpenMP threads (%d logical CPUs visible)"
                                  % (reps, pb.n_patt, ncores, os.cpu_count() or 0)}
    # the reference program itself, when its built binary travelled with the repository: that is the baseline then, and the
    # port's single-thread figure stays beside it
    ref = reference_binary_rate(pb)
    if ref is not None:
        port = {k: one[k] for k in ("value", "unit", "cores", "sample")}
        one.update(ref)
        one["kind"] = "reference"
        one["port_1core"] = port
    return one


if __name__ == "__main__":
    main()
