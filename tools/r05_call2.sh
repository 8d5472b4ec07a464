#!/bin/bash
# round 5, GPU call 2: the fixes of call 1 + the cooperative per-tree kernel (tests, per-kernel durations of the small cases, bench)
O=gpurun_out/r05b; mkdir -p $O; cd /root/repo
export PAML_AMD_JIT_SYNC=1
timeout 900 python -m pytest tests/test_engine_gpu.py -x -q -m gpu -k "keep_partials or dirty_evaluation or cooperative or small_20_state" > $O/t_engine.log 2>&1; echo engine rc=$?
timeout 300 python -m pytest tests/test_eigen_gpu.py -x -q -m gpu -k converge > $O/t_eigen.log 2>&1; echo eigen rc=$?
unset PAML_AMD_JIT_SYNC
timeout 900 python -m pytest tests/test_reference_binding_gpu.py -q -m gpu -s -k "rate_ancestor or batched_gradient or fast" > $O/t_ref.log 2>&1; echo ref rc=$?
export PAML_AMD_JIT_SYNC=1
for c in hiv_m0 hiv_m8 stewart; do
  timeout 120 python tools/small_timeline.py $c 300 > $O/tl_$c.txt 2>&1
  (cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_$c -o p -- python /root/repo/tools/small_timeline.py $c 300 > /dev/null 2>&1; cp /tmp/prof_$c/*kernel_stats.csv /root/repo/$O/stats_$c.csv 2>/dev/null || find /tmp/prof_$c -name "*kernel_stats.csv" -exec cp {} /root/repo/$O/stats_$c.csv \; )
done
PAML_AMD_COOPJIT=0 timeout 120 python tools/small_timeline.py hiv_m0 300 > $O/tl_hiv_m0_interp.txt 2>&1
unset PAML_AMD_JIT_SYNC
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo bench rc=$?
cat $O/tl_*.txt; tail -n 3 $O/t_engine.log $O/t_eigen.log; tail -n 12 $O/t_ref.log
