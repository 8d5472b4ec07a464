#!/bin/bash
export TMPDIR=/tmp
for c in stewart; do PAML_AMD_COOP=0 python tools/small_timeline.py $c 300 2>&1 | tail -1; python tools/small_timeline.py $c 300 2>&1 | tail -1; done
timeout 900 python -m pytest tests/test_engine_gpu.py tests/test_host_c.py -x -q -m gpu -k "20 or aa or stewart or lg or wag or amino or golden" 2>&1 | grep -E "passed|failed|Error" | tail -4
