#!/bin/bash
# round 4, second session: the tip-table rows without the XOR swizzle (variant library built with -DTIP_SWZ_OFF=1) against the default,
# 61 states (bench headline, back to back) and 20 states (tools/m20_probe.py)
export TMPDIR=/tmp
python -c "import torch" 2>/dev/null
for lib in "" $PWD/paml_amd/lib/exp_noswz/libpaml_amd.so "" $PWD/paml_amd/lib/exp_noswz/libpaml_amd.so; do
  echo -n "lib=${lib##*/lib/} : "
  PAML_AMD_CSRC=$PWD/paml_amd/csrc PAML_AMD_LIB=$lib python bench.py --steps 50 --warmup 10 --no-extras --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); print('ms_per_step %.4f kernel_ms %.4f frac %.4f lnL %r' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['lnL']))
except Exception as e: print('failed', e)"
done
