#!/bin/bash
# HBM traffic of the 4-state kernel on BASELINE configs[1] (32 taxa x 1e5 nucleotide patterns, GTR+G4) and at 4e6 patterns:
# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes  ->  gpurun_out/c2_pmc.txt
out=$PWD/gpurun_out
mkdir -p $out
export TMPDIR=/tmp
B="python $PWD/tools/valu_probe.py"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/c2_stats -o s -- $B > $out/c2_probe.txt 2>/dev/null
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/c2_fetch -o f -- $B > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/c2_write -o w -- $B > /dev/null 2>&1
cd - > /dev/null
{ cat $out/c2_probe.txt; find $out/c2_stats -name "*kernel_stats.csv" -exec head -6 {} \; ; python tools/pmc_summary.py $out/c2_fetch; python tools/pmc_summary.py $out/c2_write; } > $out/c2_pmc.txt 2>&1
rm -rf $out/c2_stats $out/c2_fetch $out/c2_write
cat $out/c2_pmc.txt
