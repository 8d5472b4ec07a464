#!/bin/bash
# round 5, GPU call 9: why the half-mode kernel (> 207 tips) is slow — the full build alone, under the kernel trace
O=gpurun_out/r05j; mkdir -p $O; cd /root/repo; R=/root/repo
export PAML_AMD_JIT_SYNC=1 PAML_AMD_JIT_CACHE=/tmp/jc
timeout 400 python tools/big_tree_sweep.py 230 2>&1 | grep taxa > $O/sync_230.txt
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/bt && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bt -o b -- python $R/tools/big_tree_sweep.py 230 > /tmp/bt.log 2>&1; find /tmp/bt -name "*kernel_stats.csv" -exec cp {} $R/$O/stats_230.csv \;)
cat $O/sync_230.txt; cut -d, -f1-4 $O/stats_230.csv | head -8
