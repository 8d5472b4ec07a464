#!/bin/bash
out=$PWD/gpurun_out/r04b
mkdir -p $out; export TMPDIR=/tmp
for c in hiv_m0 hiv_m8 stewart brown; do python tools/small_timeline.py $c 300 2>&1 | tail -1; done | tee $out/small_after.txt
timeout 1500 python -m pytest tests -x -q -m gpu > $out/gpu_tests.txt 2>&1
grep -E "passed|failed" $out/gpu_tests.txt | tail -2; grep -B30 "Error" $out/gpu_tests.txt | head -60
