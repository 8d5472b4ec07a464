#!/bin/bash
# round 5, GPU call 6: which of the big-tree compile flags costs the kernel its speed; where an M8 search's GPU time goes
O=gpurun_out/r05f; mkdir -p $O; cd /root/repo; R=/root/repo
for f in "-mllvm -amdgpu-load-store-vectorizer=0" "-mllvm -amdgpu-load-store-vectorizer=0 -mllvm -disable-copyprop" "-mllvm -amdgpu-load-store-vectorizer=0 -mllvm -disable-machine-cse"; do
  echo "# PAML_AMD_JIT_BIG_FLAGS=$f"; PAML_AMD_JIT_BIG_FLAGS="$f" timeout 200 python tools/big_tree_compile_probe.py 192 2>&1 | tail -1
done > $O/big_tree_flags.txt
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/mp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/mp -o m -- python $R/tools/mle_profile.py > $R/$O/mle_profile.txt 2>&1; find /tmp/mp -name "*kernel_stats.csv" -exec cp {} $R/$O/mle_m8_kernel_stats.csv \;)
python tools/time_to_mle.py > $O/time_to_mle.txt 2>&1
cat $O/big_tree_flags.txt; tail -2 $O/mle_profile.txt; cut -d, -f1-4,8 $O/mle_m8_kernel_stats.csv | head -12; cat $O/time_to_mle.txt
