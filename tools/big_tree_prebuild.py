#!/usr/bin/env python3
"""Compile the per-tree kernel of tools/big_tree_*.py's synthetic trees here (hiprtc needs no GPU) into the library's lib/jit directory, under
the environment's generator switches: the GPU box then loads instead of compiling.  usage: python tools/big_tree_prebuild.py taxa ..."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO]
from paml_amd import engine, synth  # noqa: E402
for taxa in [int(a) for a in sys.argv[1:]]:
    pb = synth.codon_m0_problem(n_tips=taxa, n_patt=1024, seed=taxa)
    t0 = time.perf_counter()
    engine.jit_prebuild(pb.tree, 61, 61, K=1, n_patt_global=65536)
    print("%d taxa: compiled in %.1f s (PAML_AMD_JIT_SPLIT=%s, quick=%s)" % (taxa, time.perf_counter() - t0, os.environ.get("PAML_AMD_JIT_SPLIT", "default"),
                                                                          os.environ.get("PAML_AMD_PREBUILD_QUICK", "0")), flush=True)
