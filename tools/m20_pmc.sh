#!/bin/bash
# counters of the 20-state kernel on one case of tools/m20_probe.py (default 0 = 32 taxa x 1e5 patterns x 4 classes)
case=${1:-0}
out=$PWD/gpurun_out/r2m20; mkdir -p $out; export TMPDIR=/tmp M20_NOCHECK=1
C="python $PWD/tools/m20_probe.py $case"
cd /tmp
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $out/p1 -o a -- $C > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $out/p2 -o b -- $C > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_LDS_ADDR_CONFLICT --kernel-trace --output-format csv -d $out/p3 -o c -- $C > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_WAIT_IFETCH --kernel-trace --output-format csv -d $out/p4 -o d -- $C > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_SMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_WAVES SQ_LDS_UNALIGNED_STALL --kernel-trace --output-format csv -d $out/p5 -o e -- $C > /dev/null 2>&1
cd - >/dev/null
for p in p1 p2 p3 p4 p5; do python tools/pmc_summary.py $out/$p | grep prune_jit; done
rm -rf $out/p1 $out/p2 $out/p3 $out/p4 $out/p5
