#!/bin/bash
# A/B bench of engine builds: tools/ab.sh lib1.so lib2.so ...   ("" = default build)
for lib in "$@"; do
  PAML_AMD_LIB=$lib python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-40s' % '${lib##*/}', 'kernel_ms=%.3f' % d['roofline']['kernel_ms'], 'step_ms=%.3f' % d['ms_per_step'], 'frac=%.3f' % d['roofline']['frac'], d['config']['kernel'], 'lnL', d['lnL'])"
done
