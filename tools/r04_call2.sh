#!/bin/bash
out=$PWD/gpurun_out/r04
mkdir -p $out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_engine_gpu.py -x -q -k "eval_branch or branch_local or branch_labels or one_rank_rccl" > $out/t_branch.txt 2>&1
tail -15 $out/t_branch.txt
timeout 300 python tools/branch_probe.py > $out/branch_probe_eig.json 2>$out/branch_probe_eig.err; cat $out/branch_probe_eig.json; tail -3 $out/branch_probe_eig.err
R=$PWD
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/branch_stats -o s -- python $R/tools/branch_probe.py > $out/branch_probe_rocprof.json 2>$out/branch_stats.err
cd - >/dev/null
find $out/branch_stats -name "*kernel_stats.csv" -exec cp {} $out/branch_eig_kernel_stats.csv \;
rm -rf $out/branch_stats
head -12 $out/branch_eig_kernel_stats.csv
cat $out/branch_eig_kernel_stats.csv | cut -c1-170 | head -24
