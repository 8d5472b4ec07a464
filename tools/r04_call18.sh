#!/bin/bash
out=$PWD/gpurun_out/r04b
mkdir -p $out; export TMPDIR=/tmp; cd /tmp
for c in hiv_m0 hiv_m8 stewart; do
  rm -rf /tmp/tr_$c
  python $GRAFT_REPO_ROOT/tools/small_timeline.py $c 200 2>&1 | tail -1
  rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$c -o t -- python $GRAFT_REPO_ROOT/tools/small_timeline.py $c 200 > /tmp/tr_$c.log 2>&1
  tail -1 /tmp/tr_$c.log
  python $GRAFT_REPO_ROOT/tools/small_timeline_digest.py /tmp/tr_$c
done 2>&1 | tee $out/small_timeline.txt
