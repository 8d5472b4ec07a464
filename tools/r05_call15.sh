#!/bin/bash
# round 5, GPU call 15: which phase of pmat_mfma_kernel costs what at 1012 workgroups (HIV M8) and at 92 (M0): compile-time variants
O=gpurun_out/r05p; mkdir -p $O; R=$PWD
for v in 0 1 2 4 8 16 31; do
  L=$R/paml_amd/lib/exp/libpaml_amd_pv$v.so; [ $v = 0 ] && L=$R/paml_amd/lib/libpaml_amd.so
  for c in hiv_m8 hiv_m0; do
    echo "== PMAT_VARIANT=$v $c"
    (cd /tmp && rm -rf /tmp/tr_$c && PAML_AMD_LIB=$L timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$c -o t -- python $R/tools/small_timeline.py $c 200 > /tmp/tr_$c.log 2>&1; python $R/tools/small_timeline_digest.py /tmp/tr_$c | head -3 | tail -1)
  done
done > $O/pmat_variants.txt 2>&1
cat $O/pmat_variants.txt
