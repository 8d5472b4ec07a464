#!/bin/bash
# run bench.py against each ablation build of the engine (kernel experiments; results are NOT parity-valid)
for v in "" NO_TIPLOAD NO_MFMA NO_STAGE NO_BARRIER; do
  if [ -z "$v" ]; then lib=""; else lib="$PWD/paml_amd/lib/abl_$v.so"; fi
  PAML_AMD_LIB=$lib python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('${v:-BASE}', 'kernel_ms=%.3f' % d['roofline']['kernel_ms'], 'step_ms=%.3f' % d['ms_per_step'], 'lnL', d['lnL'])"
done
