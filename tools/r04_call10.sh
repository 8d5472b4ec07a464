#!/bin/bash
out=$PWD/gpurun_out/r04; mkdir -p $out
export TMPDIR=/tmp PROBE_NOCHECK=1
R=$PWD
cd /tmp
for v in base "PAML_AMD_BEIG_NOMFMA=1" "PAML_AMD_BEIG_NOSTORE=1" "PAML_AMD_BEIG_NOMFMA=1 PAML_AMD_BEIG_NOSTORE=1 PAML_AMD_BEIG_NOFEVAL=1"; do
  rm -rf /tmp/bs; env $( [ "$v" != base ] && echo $v ) timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bs -o s -- python $R/tools/branch_probe.py > /tmp/bp.json 2>/tmp/bs.err
  echo "== $v"; find /tmp/bs -name "*kernel_stats.csv" -exec cat {} \; | grep "branch_eig" | sed 's/paml_amd:://g; s/(BranchEigArgs)//; s/"void branch_eig_kernel//' | awk -F'",' '{split($2,a,","); printf "%s  calls %s avg_us %.1f\n", $1, a[1], a[3]/1000}'
done > $out/branch_ablate2.txt 2>&1
cat $out/branch_ablate2.txt
