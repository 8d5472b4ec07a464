#!/bin/bash
out=$PWD/gpurun_out/r04
mkdir -p $out
export TMPDIR=/tmp
timeout 1700 python -m pytest tests/test_reference_binding_gpu.py -x -q -s > $out/t_binding.txt 2>&1
grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" $out/t_binding.txt | tail -40
