#!/bin/bash
# round 4, GPU call 1: the device-mode collective stand-in (test + measurement) and a kernel trace of the branch-local evaluation
out=$PWD/gpurun_out/r04
mkdir -p $out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_multirank_gpu.py -x -q -k "needs_a_cu or overlapping" > $out/t_multirank.txt 2>&1
tail -3 $out/t_multirank.txt
PAML_AMD_RCCL_LIB=$PWD/tests/shim/librccl_shim.so timeout 900 python tools/comm_emulated.py --steps 200 > $out/comm_emulated.txt 2>$out/comm_emulated.err
tail -60 $out/comm_emulated.txt
R=$PWD
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/branch_stats -o s -- python $R/tools/branch_probe.py > $out/branch_probe.json 2>$out/branch_stats.err
cd - >/dev/null
find $out/branch_stats -name "*kernel_stats.csv" -exec cp {} $out/branch_kernel_stats.csv \;
rm -rf $out/branch_stats
cat $out/branch_probe.json; head -12 $out/branch_kernel_stats.csv
