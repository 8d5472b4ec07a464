#!/bin/bash
# round 5, GPU call 14: P(t) on the matrix cores with several nodes per workgroup: parity with the loop forced everywhere, then timelines
O=gpurun_out/r05o; mkdir -p $O; R=$PWD
PAML_AMD_PMAT_MFMA_NPB=3 timeout 900 python -m pytest tests/test_engine_gpu.py tests/test_eigen_gpu.py -x -q > $O/t_npb3.log 2>&1; echo "npb3 tests rc=$?"; tail -3 $O/t_npb3.log
for npb in 1 0 2 4; do
  for c in hiv_m8 hiv_m0; do
    echo "== PAML_AMD_PMAT_MFMA_NPB=$npb $c"
    PAML_AMD_PMAT_MFMA_NPB=$npb timeout 120 python tools/small_timeline.py $c 300 2>&1 | tail -1
    (cd /tmp && rm -rf /tmp/tr_$c && PAML_AMD_PMAT_MFMA_NPB=$npb timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$c -o t -- python $R/tools/small_timeline.py $c 200 > /tmp/tr_$c.log 2>&1; python $R/tools/small_timeline_digest.py /tmp/tr_$c | head -4 | tail -2)
  done
done > $O/timelines.txt 2>&1
cat $O/timelines.txt
for npb in 1 0; do echo "== PAML_AMD_PMAT_MFMA_NPB=$npb"; PAML_AMD_PMAT_MFMA_NPB=$npb timeout 300 python tools/time_to_mle.py 2>&1 | grep -v Warn | tail -14; done > $O/time_to_mle.txt 2>&1
cat $O/time_to_mle.txt
