"""The exchange step against a collective that BEHAVES like RCCL's, measured on ONE GPU (profiles/r04_comm_emulated.txt).

The stand-in's device mode (tests/shim/rccl_shim.cpp, PAML_AMD_SHIM_DEVICE_US=T): ncclAllReduce returns at once and launches a
kernel on the engine's collective stream that needs a CU of its own (512 threads, 64 KB of LDS) and holds it for T microseconds —
the peer round trips of an 8-rank all-reduce of a few KB — before it delivers the sum.  For the shard a rank holds at N-way strong
scaling of the headline workload (10^6 / N codon patterns, 16 taxa, M0) the evaluation loop of bench.py is timed plain and inside
such a one-rank communicator, with two pruning streams (the default) and with one; the engine's own exchange statistics
(paml_amd_comm_stats) say where the time went.
usage: PAML_AMD_RCCL_LIB=tests/shim/librccl_shim.so python tools/comm_emulated.py [--steps 200] [--T 0,10,20,40] [--wgs 1]"""
import argparse
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from paml_amd import distributed, engine, synth  # noqa: E402


def loop(eng, branch, steps):
    d = torch.zeros(steps + 80, dtype=torch.float64, device="cuda")
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    for i in range(10):
        eng.eval_device(branch, d.data_ptr() + 8 * i)
    eng.flush()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        eng.eval_device(branch, d.data_ptr() + 8 * (10 + i))
    eng.flush()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    vals = d[:10 + steps].cpu().numpy()
    ok = bool((vals == vals[0]).all())
    # the exchange statistics of a further 64 evaluations (timed events: not part of the timing above)
    eng.comm_stats(True)
    for i in range(64):
        eng.eval_device(branch, d.data_ptr() + 8 * (10 + steps + i))
    eng.flush()
    torch.cuda.synchronize()
    st = eng.comm_stats(False, read=True)
    return ms, ok, st


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--patterns", type=int, default=1_000_000)
    ap.add_argument("--ranks", default="1,2,4,8")
    ap.add_argument("--T", default="0,10,20,40")
    ap.add_argument("--wgs", default="1")
    ap.add_argument("--modes", default="dual,single")
    ap.add_argument("--extra-env", default="", help="NAME=VALUE,... set for every engine (experiments)")
    args = ap.parse_args()
    for kv in [v for v in args.extra_env.split(",") if v]:
        k, v = kv.split("=")
        os.environ[k] = v
    full = synth.codon_m0_problem(n_tips=16, n_patt=args.patterns, estimate_pi=True)
    rows = []
    for N in [int(v) for v in args.ranks.split(",")]:
        lo, hi = distributed.shard_bounds(full.n_patt, N, 0)
        pb = full.slice_patterns(lo, hi) if N > 1 else full
        for mode in args.modes.split(","):
            os.environ["PAML_AMD_DUAL"] = "1" if mode == "dual" else "0"
            cfgs = [("plain", None, 1)] + [("T=%s" % t, int(t), int(w)) for t in args.T.split(",") for w in args.wgs.split(",")]
            for name, T, wgs in cfgs:
                if T is None:
                    os.environ.pop("PAML_AMD_SHIM_DEVICE_US", None)
                else:
                    os.environ["PAML_AMD_SHIM_DEVICE_US"] = str(T)
                    os.environ["PAML_AMD_SHIM_WGS"] = str(wgs)
                eng = engine.engine_for(pb)
                eng.comm_init(0, 1, engine.comm_unique_id() if T is not None else None, pb.n_patt, 0)
                ms, ok, st = loop(eng, pb.tree.branch, args.steps)
                row = dict(N=N, patterns=hi - lo, mode=mode, comm=name, wgs=wgs, ms=ms, same_bits=ok, **st)
                rows.append(row)
                print(json.dumps(row), flush=True)
                eng.close()
    print("\n N  patterns  mode    comm    wgs   ms/eval   vs plain   exchange_us (mean / max)   lane_wait_us (mean / max)")
    base = {}
    for r in rows:
        if r["comm"] == "plain":
            base[(r["N"], r["mode"])] = r["ms"]
        b = base[(r["N"], r["mode"])]
        print("%2d  %8d  %-6s  %-6s  %d   %8.4f  %+7.2f%%   %8.1f / %-8.1f      %8.1f / %-8.1f %s"
              % (r["N"], r["patterns"], r["mode"], r["comm"], r["wgs"], r["ms"], 100 * (r["ms"] / b - 1), r["exchange_us"], r["exchange_us_max"],
                 r["lane_wait_us"], r["lane_wait_us_max"], "" if r["same_bits"] else "MISMATCH"))
    one = base.get((1, "dual"))
    if one:
        print("\npredicted strong scaling of bench.py (10^6 patterns over N GPUs), two pruning streams: time(N = 1, plain) / time(shard, T)")
        for r in rows:
            if r["mode"] == "dual" and r["comm"] != "plain":
                print("  N = %d  %-6s wgs %d: %.2fx" % (r["N"], r["comm"], r["wgs"], one / r["ms"]))


if __name__ == "__main__":
    main()
