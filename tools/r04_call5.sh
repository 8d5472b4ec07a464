#!/bin/bash
out=$PWD/gpurun_out/r04
mkdir -p $out
export TMPDIR=/tmp
R=$PWD
cd /tmp
for v in base PAML_AMD_BEIG_NOFEVAL PAML_AMD_BEIG_NOSTORE; do
  if [ $v != base ]; then export $v=1; fi
  rm -rf /tmp/bs; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bs -o s -- python $R/tools/branch_probe.py > /tmp/bp.json 2>/tmp/bs.err
  echo "== $v"; find /tmp/bs -name "*kernel_stats.csv" -exec cat {} \; | grep "branch_eig\|branch_poly" | cut -d, -f1-4,6,7 | sed 's/paml_amd:://g; s/(paml_amd::BranchEigArgs)//'
  if [ $v != base ]; then unset $v; fi
done > $out/branch_ablate.txt 2>&1
cat $out/branch_ablate.txt
