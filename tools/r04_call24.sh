#!/bin/bash
out=$PWD/gpurun_out/r04b
mkdir -p $out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu > $out/gpu_tests.txt 2>&1
grep -E "passed|failed" $out/gpu_tests.txt | tail -2; grep -B30 "Error" $out/gpu_tests.txt | head -60
