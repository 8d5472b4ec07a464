#!/bin/bash
# (the ablation variables are read by a library built with -DPAML_AMD_JIT_EXPERIMENTS only: PAML_AMD_LIB=<dir>/libpaml_amd.so PAML_AMD_EXTRA_FLAGS=-DPAML_AMD_JIT_EXPERIMENTS python -c "from paml_amd import engine; engine.build()")
# 61-state kernel, timing only (results are garbage without the barriers): do the two waves of a SIMD gain from running half a
# product apart?  no barriers, waves 4-7 delayed at the start by SKEW x 64 cycles (a product = 60 MFMAs = 3840 cycles per wave)
for v in "X=1" "PAML_AMD_JIT_ABL_NOBAR=1" "PAML_AMD_JIT_ABL_NOBAR=1 PAML_AMD_JIT_ABL_SKEW=30" "PAML_AMD_JIT_ABL_NOBAR=1 PAML_AMD_JIT_ABL_SKEW=60" "PAML_AMD_JIT_ABL_NOBAR=1 PAML_AMD_JIT_ABL_SKEW=120" "PAML_AMD_JIT_ABL_SKEW=60"; do
  echo "== $v"; env $v ABL_TAG="$v" PAML_AMD_JIT_CACHE=0 python tools/jit_ablate.py 2>/dev/null | tail -1
done
