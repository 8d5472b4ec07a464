#!/bin/bash
O=gpurun_out/r05r; mkdir -p $O; R=$PWD
for v in base nt; do
  L=$R/paml_amd/lib/exp/libpaml_amd_$v.so; [ $v = base ] && L=$R/paml_amd/lib/libpaml_amd.so
  for c in hiv_m8 hiv_m0; do
    echo "== $v $c"
    PAML_AMD_LIB=$L timeout 120 python tools/small_timeline.py $c 300 2>&1 | tail -1
    (cd /tmp && rm -rf /tmp/tr_$c && PAML_AMD_LIB=$L timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$c -o t -- python $R/tools/small_timeline.py $c 200 > /tmp/tr_$c.log 2>&1; python $R/tools/small_timeline_digest.py /tmp/tr_$c | head -4 | tail -2)
  done
done > $O/pmat_nt.txt 2>&1
cat $O/pmat_nt.txt
