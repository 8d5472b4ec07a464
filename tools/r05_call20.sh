#!/bin/bash
# round 5, GPU call 20: does the headline kernel (57 ops, 27 KB of code) care about block splits?  (it does not get them by default)
O=gpurun_out/r05u; mkdir -p $O
for v in "" "PAML_AMD_JIT_SPLIT=br PAML_AMD_JIT_SPLIT_EVERY=8" "PAML_AMD_JIT_SPLIT=br PAML_AMD_JIT_SPLIT_EVERY=4" "PAML_AMD_JIT_SPLIT=asm PAML_AMD_JIT_SPLIT_EVERY=4" ""; do
  echo "== [$v]"; env $v timeout 200 python bench.py --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step %.4f  kernel_ms %.4f  frac %.4f' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac']))"
done > $O/headline_split.txt 2>&1
cat $O/headline_split.txt
