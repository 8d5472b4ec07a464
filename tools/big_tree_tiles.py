#!/usr/bin/env python3
"""Workgroup timeline (PAML_AMD_PROF_TILES) of the per-tree kernel of a large tree.  usage: python tools/big_tree_tiles.py taxa [npatt]"""
import os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO]
taxa = int(sys.argv[1]); npatt = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
dump = "/tmp/tiles_%d.bin" % taxa
os.environ.update(PAML_AMD_PROF_OPS=dump, PAML_AMD_PROF_TILES="1", PAML_AMD_JIT="1")
import torch  # noqa: E402
from paml_amd import engine, synth  # noqa: E402
pb = synth.codon_m0_problem(n_tips=taxa, n_patt=npatt, seed=taxa)
eng = engine.engine_for(pb)
for _ in range(3):
    v = eng.eval(pb.tree.branch)["lnL"]
print(taxa, eng.kernel_name, v, flush=True)
eng.close()
print(subprocess.run([sys.executable, os.path.join(REPO, "tools", "prof_tiles.py"), dump], capture_output=True, text=True).stdout[:5000])
