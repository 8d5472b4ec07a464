#!/bin/bash
# 20-state kernel: static tiles / per-wave tickets x equal / unequal wave priorities (jit.h jit_generate_m20); lnL must not move
for v in "PAML_AMD_M20_STATIC=1 PAML_AMD_M20_NOPRIO=1" "PAML_AMD_M20_STATIC=1" "PAML_AMD_M20_NOPRIO=1" "X=1"; do echo "== $v"; env $v PAML_AMD_JIT_CACHE=0 python tools/m20_probe.py 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('   %-50s prune %.4f ms  %.1f TF  ms/eval %.4f  lnL %s  slice %.1e %.1e' % (d['case'], d['ms_prune'], d['tflops'], d['ms_per_eval'], float(d['lnL']).hex(), d['slice_rel_diff'], d['slice_max_lnf_diff']))"; done
