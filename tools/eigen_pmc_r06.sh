#!/bin/bash
# Round 6: counters of the eigen kernel (tools/eigen_probe.py's launches), each group in its own --pmc pass with --kernel-trace only.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
REPO=$(pwd); out=$REPO/gpurun_out/eigpmc; mkdir -p $out
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_SALU"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-60)
  (cd /tmp && timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/eigpmc_$tag -o p -- python $REPO/tools/eigen_probe.py > /tmp/eigpmc_$tag.log 2>&1) || tail -3 /tmp/eigpmc_$tag.log
  python tools/pmc_summary.py /tmp/eigpmc_$tag | grep -i eigen
done | tee $out/pmc_summary.txt
