#!/bin/bash
# round 5, GPU call 3: resolved P(t) table + no row-major copy, the keep-partials fix, per-kernel durations of the small cases, the write-only bound
O=gpurun_out/r05c; mkdir -p $O; cd /root/repo; R=/root/repo
tools/hbm_write_peak > $O/hbm_write_peak.txt 2>&1
export PAML_AMD_JIT_SYNC=1
timeout 900 python -m pytest tests/test_engine_gpu.py -x -q -m gpu -k "keep_partials or dirty_evaluation or cooperative or small_20_state or pmat or golden or closed_form or unrest" > $O/t_engine.log 2>&1; echo engine rc=$?
for c in hiv_m0 hiv_m8 stewart brown; do
  timeout 120 python tools/small_timeline.py $c 300 2>&1 | tail -1 > $O/tl_$c.txt
  (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/tr_$c && timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$c -o t -- python $R/tools/small_timeline.py $c 200 > /tmp/tr_$c.log 2>&1; python $R/tools/small_timeline_digest.py /tmp/tr_$c | head -6) >> $O/tl_$c.txt 2>&1
done
unset PAML_AMD_JIT_SYNC
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo bench rc=$?
cat $O/hbm_write_peak.txt $O/tl_*.txt; tail -n 3 $O/t_engine.log
