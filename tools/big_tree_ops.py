#!/usr/bin/env python3
"""Per-op cycle stamps of the per-tree kernel of a large tree (PAML_AMD_PROF_OPS build of the generated kernel): where a walk's time goes.
usage: python tools/big_tree_ops.py taxa   -> prints the digest of tools/prof_ops.py"""
import os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO]
taxa = int(sys.argv[1])
dump = "/tmp/ops_%d.bin" % taxa
os.environ.update(PAML_AMD_PROF_OPS=dump, PAML_AMD_JIT="1", PAML_AMD_JIT_CACHE="0")
import torch  # noqa: E402
from paml_amd import engine, synth  # noqa: E402
pb = synth.codon_m0_problem(n_tips=taxa, n_patt=128 * 256, seed=taxa)
eng = engine.engine_for(pb)
for _ in range(3):
    v = eng.eval(pb.tree.branch)["lnL"]
eng.profile(True)
for _ in range(3):
    eng.eval(pb.tree.branch)
pr = eng.profile_read(); eng.profile(False)
print(taxa, eng.kernel_name, v, "pruning kernel %.3f ms per launch (HIP events), %d tiles" % (pr["ms_prune"] / max(1, pr["n_evals"]), (pb.n_patt + 127) // 128), flush=True)
eng.close()
print(subprocess.run([sys.executable, os.path.join(REPO, "tools", "prof_ops.py"), dump], capture_output=True, text=True).stdout[:6000])
