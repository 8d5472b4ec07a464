#!/bin/bash
# large trees: the block splits of the per-tree kernel (prebuilt here into lib/jit: no compile on the box)
mkdir -p gpurun_out/r05l
O=gpurun_out/r05l/split.txt
: > $O
run() { echo "== $*" >> $O; env "$@" timeout 170 python tools/big_tree_npatt.py $TAXA 65536 2>&1 | grep -v Warning | tail -2 >> $O; }
TAXA=230
run PAML_AMD_JIT_SPLIT=br
run PAML_AMD_JIT_SPLIT=asm
run PAML_AMD_JIT_SPLIT=br PAML_AMD_JIT_SPLIT_EVERY=2
run PAML_AMD_JIT_SPLIT=br PAML_AMD_JIT_SPLIT_EVERY=4
run PAML_AMD_JIT_SPLIT=br PAML_AMD_JIT_SPLIT_EVERY=16
TAXA=192
run PAML_AMD_JIT_SPLIT=br
run PAML_AMD_JIT_SPLIT=br PAML_AMD_JIT_SPLIT_EVERY=4
run PAML_AMD_JIT_SPLIT=br PAML_AMD_JIT_SPLIT_EVERY=16
TAXA=400
run PAML_AMD_JIT_SPLIT=br
TAXA=96
run PAML_AMD_JIT_SPLIT=br
TAXA=49
run PAML_AMD_JIT_SPLIT=br
cat $O
