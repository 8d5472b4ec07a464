#!/usr/bin/env python3
"""bench.py's branch block alone (eval_branch at the headline size: first call, refill by the interpreter and by the forest's per-tree
kernel, walk, hit, forming kernel).  usage: python tools/refill_probe.py [patterns]"""
import json, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO]
import bench
from paml_amd import engine, synth
npatt = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
pb = synth.codon_m0_problem(n_tips=16, n_patt=npatt)
eng = engine.engine_for(pb)
lnl = eng.eval(pb.tree.branch)["lnL"]
eng.close()
print(json.dumps(bench.bench_branch(engine, pb, lnl), indent=1))
