#!/bin/bash
# 61-state per-tree kernel: timing ablations (results of the NOBAR variants are garbage, only the time counts)
run() { echo -n "$1: "; env $1 PAML_AMD_BENCH_NOCHECK=1 python bench.py --steps 50 --warmup 10 --no-extras --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); print('ms_per_step %.4f kernel_ms %.4f frac %.4f lnL %r' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['lnL']))
except Exception as e: print('failed', e)"; }
run X=1
for v in "$@"; do run "$v"; done
run X=1
