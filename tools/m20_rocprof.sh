#!/bin/bash
# kernel-trace durations of the 20-state kernel (rocprofv3's dispatch begin / end) beside the HIP-event figure of tools/m20_probe.py;
# the trace holds the synchronous loops (lower clock) and the back-to-back loop: the minimum is the production figure
out=$PWD/gpurun_out/m20rp; mkdir -p $out; export TMPDIR=/tmp M20_NOCHECK=1
for c in 0 1; do
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out/s$c -o s -- python $OLDPWD/tools/m20_probe.py $c 2>/dev/null | tail -1)
  find $out/s$c -name "*kernel_stats.csv" -exec head -2 {} \;
done
rm -rf $out
