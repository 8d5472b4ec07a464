#!/bin/bash
# round 4, second session: split-phase synchronisation of the 61-state per-tree kernel (PAML_AMD_JIT_SPLITBAR: 2 = LDS counter where the
# barrier was, 1 = arrival behind the last operand read; PAML_AMD_JIT_SKEW = initial offset of the second wave of each SIMD, x 64 cycles)
export TMPDIR=/tmp
python -c "import torch" 2>/dev/null
run() { echo -n "$1 : "; env $1 timeout 120 python bench.py --steps 50 --warmup 10 --no-extras --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json
t=sys.stdin.read()
try:
    d=json.loads(t); print('ms_per_step %.4f kernel_ms %.4f frac %.4f lnL %r' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['lnL']))
except Exception as e: print('failed', t[-300:])"; }
for v in "$@"; do run "$v"; done
