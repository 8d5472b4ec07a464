#!/bin/bash
export TMPDIR=/tmp
for c in hiv_m0 hiv_m8; do python tools/small_timeline.py $c 300 2>&1 | tail -1; done
cd /tmp
for c in hiv_m0; do
  rm -rf /tmp/tr_$c
  rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$c -o t -- python $GRAFT_REPO_ROOT/tools/small_timeline.py $c 200 > /tmp/tr_$c.log 2>&1
  python $GRAFT_REPO_ROOT/tools/small_timeline_digest.py /tmp/tr_$c | head -6
done
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_engine_gpu.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|error" | tail -5
