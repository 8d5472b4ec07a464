#!/bin/bash
# large trees: block splits against the one-block kernel in the same run (all prebuilt into lib/jit), then the time a fresh box needs to compile
mkdir -p gpurun_out/r05m
O=gpurun_out/r05m/split_ab.txt
: > $O
run() { echo "== $*" >> $O; env "$@" timeout 170 python tools/big_tree_npatt.py $TAXA 65536 2>&1 | grep "taxa x" >> $O; }
for TAXA in 49 64 96 128 192; do
  run PAML_AMD_JIT_SPLIT=0
  run PAML_AMD_JIT_SPLIT=br
done
TAXA=300; run PAML_AMD_JIT_SPLIT=br
cat $O
timeout 300 python tools/big_tree_compile_probe.py 192 400 2>&1 | grep -v Warn | tail -4 | tee gpurun_out/r05m/compile_probe.txt
