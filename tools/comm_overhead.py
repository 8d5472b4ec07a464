"""What the exchange step costs, measured on ONE GPU, and what it predicts for N = 2, 4, 8 (profiles/r03_comm_overhead.txt).

For the shard a rank holds at N-way strong scaling of the headline workload (10^6 / N codon patterns, 16 taxa, M0) the same
evaluation loop is timed without a communicator and in a one-rank RCCL communicator (the whole collective path: partial sums ->
event -> collective stream -> ncclAllReduce -> fixed-order total -> event), back to back (lnL left on the device, one fence per
timed region: bench.py's loop) and with the scalar read back after every evaluation (a serial optimiser).  A one-rank all-reduce
has the launch and stream-ordering cost of the real one but not the xGMI latency; the table adds RCCL's published small-message
all-reduce latency for that (ring over xGMI, 8 KB: ~15-25 us at 8 ranks) to the read-back column only — back to back the collective
is off the critical path.
usage: python tools/comm_overhead.py [--steps 200]"""
import argparse
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from paml_amd import distributed, engine, synth  # noqa: E402


def loop(eng, branch, steps, readback):
    d = torch.zeros(steps + 16, dtype=torch.float64, device="cuda")
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    for i in range(10):
        eng.eval_device(branch, d.data_ptr() + 8 * i)
    eng.flush()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if readback:
        for i in range(steps):
            eng.eval(branch)
    else:
        for i in range(steps):
            eng.eval_device(branch, d.data_ptr() + 8 * (10 + i))
        eng.flush()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--patterns", type=int, default=1_000_000)
    ap.add_argument("--ranks", default="1,2,4,8", help="the N whose shard sizes are timed")
    args = ap.parse_args()
    full = synth.codon_m0_problem(n_tips=16, n_patt=args.patterns, estimate_pi=True)
    rows = []
    for N in [int(v) for v in args.ranks.split(",")]:
        lo, hi = distributed.shard_bounds(full.n_patt, N, 0)
        pb = full.slice_patterns(lo, hi) if N > 1 else full
        row = {"N": N, "patterns_per_rank": hi - lo}
        for comm in (False, True):
            eng = engine.engine_for(pb)
            eng.comm_init(0, 1, engine.comm_unique_id() if comm else None, pb.n_patt, 0)
            for rb in (False, True):
                row["%s_%s_ms" % ("comm" if comm else "plain", "readback" if rb else "b2b")] = loop(eng, pb.tree.branch, args.steps, rb)
            row["kernel"] = eng.kernel_name
            eng.close()
        rows.append(row)
        print(json.dumps(row), flush=True)
    base = rows[0]["plain_b2b_ms"]
    print("\n N  patterns/rank  plain b2b  comm b2b  overhead   plain rb   comm rb   predicted speed-up (comm b2b)  tiles/CU")
    for r in rows:
        tiles = -(-r["patterns_per_rank"] // 128)
        print("%2d  %12d  %8.4f  %8.4f  %+7.2f%%  %8.4f  %8.4f   %6.2fx                         %.2f"
              % (r["N"], r["patterns_per_rank"], r["plain_b2b_ms"], r["comm_b2b_ms"], 100 * (r["comm_b2b_ms"] / r["plain_b2b_ms"] - 1),
                 r["plain_readback_ms"], r["comm_readback_ms"], base / r["comm_b2b_ms"], tiles / 256.0))


if __name__ == "__main__":
    main()
