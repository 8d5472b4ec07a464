#!/bin/bash
# HBM traffic of the 4-state kernel on BASELINE configs[1] (32 taxa x 1e5 nucleotide patterns, GTR+G4) and at 4e6 patterns:
# rocprofv3 kernel stats, then --pmc FETCH_SIZE / WRITE_SIZE in separate passes  ->  gpurun_out/c2_pmc.txt
out=$PWD/gpurun_out
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
for c in c2 c2big; do
  B="python $OLDPWD/tools/valu_probe.py $c"
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/${c}_stats -o s -- $B > $out/${c}_probe.txt 2>/dev/null
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/${c}_fetch -o f -- $B > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/${c}_write -o w -- $B > /dev/null 2>&1
done
cd - > /dev/null
for c in c2 c2big; do
  echo "== $c"; cat $out/${c}_probe.txt; find $out/${c}_stats -name "*kernel_stats.csv" -exec head -4 {} \;
  python tools/pmc_summary.py $out/${c}_fetch | grep prune; python tools/pmc_summary.py $out/${c}_write | grep prune
  rm -rf $out/${c}_stats $out/${c}_fetch $out/${c}_write $out/${c}_probe.txt
done > $out/c2_pmc.txt 2>&1
cat $out/c2_pmc.txt
