#!/bin/bash
# round 5, GPU call 13: the NEB / BEB evaluations of the patched reference on the engine: the parity test, then where the time goes
O=gpurun_out/r05n; mkdir -p $O; R=/root/repo
timeout 600 python -m pytest tests/test_reference_binding_gpu.py -x -q -s -k "neb_and_beb" 2>&1 | tail -15 > $O/t_beb.log
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
  for e in "" "PAML_AMD_NO_BEB=1"; do
    s=$(date +%s%N); env PAML_AMD_TIMING=1 $e $R/oracle/_ref/codeml_gpu codeml.ctl < /dev/null > out.txt 2>&1; t=$(( ($(date +%s%N) - s) / 1000000 ))
    echo "HIV NSsites 0 2 [$e] wall $t ms"; grep "paml_amd timing\|^lnL" out.txt mlc | head -4
  done
done > $R/$O/codeml_gpu_timing.txt 2>&1
sed -e "s#../data/#$R/tests/golden/data/#" -e "s#../ctl/#$R/tests/golden/ctl/#" $R/tests/golden/ctl/lyso_bsa.ctl > codeml.ctl; echo "outfile = mlc" >> codeml.ctl; echo "noisy = 0" >> codeml.ctl
for rep in 1 2; do
for e in "" "PAML_AMD_NO_BEB=1"; do
  s=$(date +%s%N); env PAML_AMD_TIMING=1 $e $R/oracle/_ref/codeml_gpu codeml.ctl < /dev/null > out.txt 2>&1; t=$(( ($(date +%s%N) - s) / 1000000 ))
  echo "branch-site A (lysozyme) [$e] wall $t ms"; grep "paml_amd timing" out.txt; grep "^lnL" mlc | head -2
done; done >> $R/$O/codeml_gpu_timing.txt 2>&1
cat $R/$O/t_beb.log $R/$O/codeml_gpu_timing.txt
