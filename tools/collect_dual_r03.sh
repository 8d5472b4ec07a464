#!/bin/bash
# Round-3 evidence for the two pruning streams (profiles/r03_dual_stream.txt), run on the GPU box through gpurun -> gpurun_out/r3dual/
out=$PWD/gpurun_out/r3dual
mkdir -p $out
export TMPDIR=/tmp
R=$PWD
flt() { grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids"; }
{ echo "# tools/comm_overhead.py, default (two pruning streams)"; python tools/comm_overhead.py 2>&1 | flt | grep -A6 "patterns/rank"
  echo "# PAML_AMD_DUAL=0 (one pruning stream)"; PAML_AMD_DUAL=0 python tools/comm_overhead.py 2>&1 | flt | grep -A6 "patterns/rank"; } > $out/comm_overhead.txt
S=8p,8c,4p,2p,1p,1c,8p
{ for l in 2 3 4; do echo "PAML_AMD_LANES=$l  $(PAML_AMD_LANES=$l python tools/dual_probe.py $S 2>&1 | grep '^[0-9]')"; done
  echo "PAML_AMD_DUAL=0    $(PAML_AMD_DUAL=0 python tools/dual_probe.py $S 2>&1 | grep '^[0-9]')"
  echo "normal priority   $(PAML_AMD_STREAM_PRIO_NORMAL=1 python tools/dual_probe.py $S 2>&1 | grep '^[0-9]')"; } > $out/lanes.txt
{ for s in 8p 1p; do echo "# hand-over, $s"; PAML_AMD_JIT_CACHE=0 PAML_AMD_PROF_TILES=1 PAML_AMD_PROF_OPS=/tmp/tl.bin python tools/dual_probe.py $s 2>&1 | grep '^[0-9]'; python tools/handover.py /tmp/tl.bin; done; } > $out/handover.txt
cd /tmp
for n in 124928 1000000; do
  rm -rf /tmp/tl; rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o tl -- python $R/bench.py --no-extras --no-cpu-baseline --steps 30 --patterns $n > /tmp/b.log 2>&1
  { echo "# $n patterns, kernel trace of bench.py's timed loop (rows 60..)"; python $R/tools/raw_timeline.py /tmp/tl 28 60; } > $out/timeline_$n.txt
done
cat $out/comm_overhead.txt $out/lanes.txt $out/handover.txt
