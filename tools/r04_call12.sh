#!/bin/bash
export TMPDIR=/tmp
R=$PWD
cd /tmp
for lib in "" "$R/tools/libpaml_amd_licm.so" "" "$R/tools/libpaml_amd_licm.so"; do
  rm -rf /tmp/bs; env PAML_AMD_LIB=$lib timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bs -o s -- python $R/tools/branch_probe.py > /tmp/bp.json 2>/tmp/bs.err
  echo "== lib=${lib:-default (machine LICM off)}"; grep "^{" /tmp/bp.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k: round(v,4) for k,v in d.items() if k.endswith('_ms')})"
  find /tmp/bs -name "*kernel_stats.csv" -exec cat {} \; | grep "branch_eig\|poly\|total" | sed 's/paml_amd:://g; s/(BranchEigArgs)//; s/"void branch_eig_kernel//' | awk -F'",' '{split($2,a,","); printf "%s  calls %s avg_us %.1f\n", $1, a[1], a[3]/1000}'
done
