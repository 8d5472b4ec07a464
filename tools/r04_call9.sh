#!/bin/bash
out=$PWD/gpurun_out/r04
mkdir -p $out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_engine_gpu.py tests/test_multirank_gpu.py -x -q -k "eval_branch or branch_local or branch_labels or one_rank_rccl or ranks_on_one_gpu or overlapping or needs_a_cu" > $out/t_branch.txt 2>&1
grep "passed\|failed" $out/t_branch.txt | tail -2; grep -B30 "Error" $out/t_branch.txt | head -60
R=$PWD
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/branch_stats -o s -- python $R/tools/branch_probe.py > $out/branch_probe_rocprof.json 2>$out/branch_stats.err
cd - >/dev/null
find $out/branch_stats -name "*kernel_stats.csv" -exec cp {} $out/branch_eig_kernel_stats.csv \;
rm -rf $out/branch_stats
grep "^{" $out/branch_probe_rocprof.json
grep "eig\|poly\|total" $out/branch_eig_kernel_stats.csv | cut -c1-170
