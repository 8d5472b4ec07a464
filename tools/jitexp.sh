#!/bin/bash
# kernel experiments: tools/jitexp.sh "ENV=1 ENV2=2" ...  -> one bench line (+ per-op cycles of wave 4) per environment set
for envs in "$@"; do
  echo "== $envs"
  env $envs python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('kernel_ms=%.3f' % d['roofline']['kernel_ms'], 'step_ms=%.3f' % d['ms_per_step'], 'frac=%.3f' % d['roofline']['frac'], d['config']['kernel'], 'lnL', d['lnL'])"
  env $envs PAML_AMD_PROF_OPS=/tmp/ops.bin PAML_AMD_PROF_TID=256 python bench.py --steps 2 --warmup 1 --no-cpu-baseline >/dev/null 2>&1; python tools/prof_ops.py /tmp/ops.bin | grep "per-op"
done
