#!/bin/bash
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_engine_gpu.py -x -q -m gpu -k "pmat or ambiguity or golden or node_scaling or underflow or eigen or branch_labels or random_models" 2>&1 | tail -5
for c in hiv_m0 hiv_m8; do
  PAML_AMD_PMAT_MFMA=0 python tools/small_timeline.py $c 300 2>&1 | tail -1
  python tools/small_timeline.py $c 300 2>&1 | tail -1
done
cd /tmp
for c in hiv_m0 hiv_m8; do
  rm -rf /tmp/tr_$c
  rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$c -o t -- python $GRAFT_REPO_ROOT/tools/small_timeline.py $c 200 > /tmp/tr_$c.log 2>&1
  python $GRAFT_REPO_ROOT/tools/small_timeline_digest.py /tmp/tr_$c | head -6
done
