#!/bin/bash
# large trees: the time a box without the code objects needs (background compile beside interpreter evaluations)
mkdir -p gpurun_out/r05m /tmp/jitsave
mv paml_amd/lib/jit/*.hsaco /tmp/jitsave/
timeout 500 python tools/big_tree_compile_probe.py 96 192 400 2>&1 | grep -v Warn | tail -3 | tee gpurun_out/r05m/compile_probe.txt
