#!/bin/bash
# Round evidence, run on the GPU box:  tools/collect_profiles.sh <tag>   -> gpurun_out/prof_<tag>/
# (kernel-trace stats, HBM traffic counters in separate --pmc passes, MFMA busy counters, the plain bench line)
tag=${1:-r01}
out=$PWD/gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
B="python $PWD/bench.py --steps 10 --warmup 2 --no-cpu-baseline"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o s -- $B > $out/bench_under_rocprof.json 2>$out/stats.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_fetch -o f -- $B > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc_write -o w -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 --kernel-trace --output-format csv -d $out/pmc_mfma -o m -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $out/pmc_lds -o l -- $B > /dev/null 2>&1
cd - > /dev/null
{
  echo "# FETCH_SIZE / WRITE_SIZE (KiB per dispatch, separate passes)"; python tools/pmc_summary.py $out/pmc_fetch; python tools/pmc_summary.py $out/pmc_write
  echo "# MFMA"; python tools/pmc_summary.py $out/pmc_mfma
  echo "# LDS"; python tools/pmc_summary.py $out/pmc_lds
} > $out/pmc_summary.txt 2>&1
find $out/stats -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats.csv \;
python bench.py > $out/bench_1gpu.json 2>$out/bench.err
# keep the merged directory small: raw traces are not needed
rm -rf $out/stats $out/pmc_fetch $out/pmc_write $out/pmc_mfma $out/pmc_lds
cat $out/kernel_stats.csv | head -8; cat $out/pmc_summary.txt; cat $out/bench_1gpu.json
