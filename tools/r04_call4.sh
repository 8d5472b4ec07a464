#!/bin/bash
out=$PWD/gpurun_out/r04
mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -x -q -m gpu > $out/gpu_tests.txt 2>&1
tail -4 $out/gpu_tests.txt
timeout 300 python tools/branch_probe.py > $out/branch_probe_eig.json 2>$out/branch_probe_eig.err; cat $out/branch_probe_eig.json; tail -3 $out/branch_probe_eig.err
R=$PWD
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/branch_stats -o s -- python $R/tools/branch_probe.py > $out/branch_probe_rocprof.json 2>$out/branch_stats.err
cd - >/dev/null
find $out/branch_stats -name "*kernel_stats.csv" -exec cp {} $out/branch_eig_kernel_stats.csv \;
rm -rf $out/branch_stats
cut -c1-150 $out/branch_eig_kernel_stats.csv | head -14
for L in 2 3; do
  echo "== PAML_AMD_LANES=$L"
  PAML_AMD_LANES=$L PAML_AMD_RCCL_LIB=$PWD/tests/shim/librccl_shim.so timeout 600 python tools/comm_emulated.py --steps 200 --ranks 8 --T 40,100 --wgs 1,2 --modes dual 2>>$out/comm_emulated.err | grep -A8 "patterns  mode"
done > $out/comm_emulated_lanes.txt
cat $out/comm_emulated_lanes.txt
