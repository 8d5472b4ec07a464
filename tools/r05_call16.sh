#!/bin/bash
# round 5, GPU call 16: pmat_mfma_kernel without the ambiguity map for plain codes, 16-byte stores: parity, then timelines
O=gpurun_out/r05q; mkdir -p $O; R=$PWD
timeout 1200 python -m pytest tests/test_engine_gpu.py tests/test_eigen_gpu.py tests/test_host_c.py -m gpu -x -q > $O/t.log 2>&1; echo "tests rc=$?"; tail -3 $O/t.log
for c in hiv_m8 hiv_m0 stewart; do
  timeout 120 python tools/small_timeline.py $c 300 2>&1 | tail -1
  (cd /tmp && rm -rf /tmp/tr_$c && timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$c -o t -- python $R/tools/small_timeline.py $c 200 > /tmp/tr_$c.log 2>&1; python $R/tools/small_timeline_digest.py /tmp/tr_$c | head -6)
done > $O/timelines.txt 2>&1
cat $O/timelines.txt
