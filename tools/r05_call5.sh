#!/bin/bash
# round 5, GPU call 5: store variants of the keep-partials kernel, where the patched reference's time goes, the big-tree compile, M8 again
O=gpurun_out/r05e; mkdir -p $O; cd /root/repo; R=/root/repo
for v in 0 1 2; do PAML_AMD_JIT_STORE=$v timeout 200 python tools/keep_probe.py 10 2>&1 | tail -1; done > $O/keep_probe.txt
# the reference's own optimiser on the engine: HIV NSsites 0 2 and branch-site A, with and without the batched gradient
mkdir -p /tmp/hv && cd /tmp/hv
cat > codeml.ctl <<CTL
seqfile = $R/tests/golden/data/HIVenvSweden.txt
treefile = $R/tests/golden/data/HIVenvSweden.trees
outfile = mlc
noisy = 0
verbose = 0
runmode = 0
seqtype = 1
CodonFreq = 2
model = 0
NSsites = 0 2
icode = 0
fix_kappa = 0
kappa = .3
fix_omega = 0
omega = 1.3
ncatG = 10
getSE = 0
RateAncestor = 0
Small_Diff = .45e-6
cleandata = 1
fix_blength = 0
CTL
for rep in 1 2; do
  for e in "" "PAML_AMD_NO_BATCH_GRADIENT=1"; do
    s=$(date +%s%N); env PAML_AMD_TIMING=1 $e $R/oracle/_ref/codeml_gpu codeml.ctl < /dev/null > out.txt 2>&1; t=$(( ($(date +%s%N) - s) / 1000000 ))
    echo "HIV NSsites 0 2 [$e] wall $t ms"; grep "paml_amd timing\|^lnL" out.txt mlc | head -4
  done
done > $R/$O/codeml_gpu_timing.txt 2>&1
sed -e "s#../data/#$R/tests/golden/data/#" -e "s#../ctl/#$R/tests/golden/ctl/#" $R/tests/golden/ctl/lyso_bsa.ctl > codeml.ctl; echo "outfile = mlc" >> codeml.ctl; echo "noisy = 0" >> codeml.ctl
for e in "" "PAML_AMD_NO_BATCH_GRADIENT=1"; do
  s=$(date +%s%N); env PAML_AMD_TIMING=1 $e $R/oracle/_ref/codeml_gpu codeml.ctl < /dev/null > out.txt 2>&1; t=$(( ($(date +%s%N) - s) / 1000000 ))
  echo "branch-site A (lysozyme) [$e] wall $t ms"; grep "paml_amd timing" out.txt; grep "^lnL" mlc | head -2
done >> $R/$O/codeml_gpu_timing.txt 2>&1
cd $R
export PAML_AMD_JIT_SYNC=1
for c in hiv_m8 hiv_m0; do
  timeout 120 python tools/small_timeline.py $c 300 2>&1 | tail -1 > $O/tl_$c.txt
  (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/tr_$c && timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$c -o t -- python $R/tools/small_timeline.py $c 200 > /tmp/tr_$c.log 2>&1; python $R/tools/small_timeline_digest.py /tmp/tr_$c | head -6) >> $O/tl_$c.txt 2>&1
done
unset PAML_AMD_JIT_SYNC
timeout 300 python tools/big_tree_compile_probe.py 96 192 > $O/big_tree.txt 2>&1; PAML_AMD_JIT_BIG_DEFAULT_FLAGS=1 timeout 300 python tools/big_tree_compile_probe.py 192 >> $O/big_tree.txt 2>&1
cat $O/keep_probe.txt $O/codeml_gpu_timing.txt $O/tl_*.txt; tail -n 12 $O/big_tree.txt
