#!/bin/bash
# round 4, second session, call 1: (a) time to the MLEs with the eigen warm start inside pamlh_optimize (cold / warm / warm + every batch on the
# device), (b) the branch-local kernels with non-temporal stores / loads (tools/build_variant.sh nt1 nt2 nt3)
out=$PWD/gpurun_out/r04b
mkdir -p $out
export TMPDIR=/tmp
python -c "import torch" 2>/dev/null
( echo "== cold (PAMLH_EIGEN_WARM=0)"; PAMLH_EIGEN_WARM=0 timeout 300 python tools/time_to_mle.py
  echo "== warm start in pamlh_optimize (default)"; timeout 300 python tools/time_to_mle.py
  echo "== warm start, every batch on the device (PAMLH_DEVICE_EIGEN_MIN=1)"; PAMLH_DEVICE_EIGEN_MIN=1 timeout 300 python tools/time_to_mle.py
  echo "== warm start, batches of >= 4 on the device (PAMLH_DEVICE_EIGEN_MIN=4)"; PAMLH_DEVICE_EIGEN_MIN=4 timeout 300 python tools/time_to_mle.py
) > $out/time_to_mle.txt 2>&1
tail -60 $out/time_to_mle.txt
for v in "" nt1 nt2 nt3; do
  lib=""; [ -n "$v" ] && lib=$PWD/paml_amd/lib/exp/libpaml_amd_$v.so
  echo "== branch_probe lib=${v:-default}"
  PAML_AMD_LIB=$lib timeout 300 python tools/branch_probe.py 2>&1 | tail -1
done > $out/branch_nt.txt 2>&1
cat $out/branch_nt.txt
