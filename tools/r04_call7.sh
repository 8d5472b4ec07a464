#!/bin/bash
out=$PWD/gpurun_out/r04
mkdir -p $out
export TMPDIR=/tmp
python __graft_entry__.py smoke 2>&1 | grep "^smoke"
timeout 900 python -m pytest tests/test_engine_gpu.py -x -q -k "golden or full_size or specialised or random_models or nssites_sweep or spills or background" > $out/t_rowtail.txt 2>&1; grep "passed\|failed" $out/t_rowtail.txt | tail -2; grep -B5 "Error" $out/t_rowtail.txt | head -30
for v in "" "PAML_AMD_JIT_NOROWTAIL=1"; do
  echo "== $v"
  env $v python bench.py --no-extras --no-cpu-baseline --steps 50 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('kernel_ms=%.4f step_ms=%.4f frac=%.4f lnL=%r' % (d['roofline']['kernel_ms'], d['ms_per_step'], d['roofline']['frac'], d['lnL']))"
done
