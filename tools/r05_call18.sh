#!/bin/bash
O=gpurun_out/r05s; mkdir -p $O; R=$PWD
for c in hiv_m8 hiv_m0 hiv_m8 hiv_m0; do
    timeout 120 python tools/small_timeline.py $c 300 2>&1 | tail -1
    (cd /tmp && rm -rf /tmp/tr_$c && timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$c -o t -- python $R/tools/small_timeline.py $c 200 > /tmp/tr_$c.log 2>&1; python $R/tools/small_timeline_digest.py /tmp/tr_$c | head -4 | tail -2)
done > $O/pmat_nt_default.txt 2>&1
cat $O/pmat_nt_default.txt
timeout 200 python tools/time_to_mle.py 2>&1 | grep "hiv_m[078]\|lyso_bsa \|ecp_cmc" | tee $O/time_to_mle.txt
timeout 100 python bench.py --no-extras --no-cpu-baseline 2>/dev/null | cut -c1-300
