#!/bin/bash
# the pamlh_lnl driver (ML estimation + SEs) on this round's host options: lnL reached and wall time per analysis
for c in "codeml hiv_fmutsel0.ctl" "codeml hiv_fmutsel_est.ctl" "codeml mtcdna_aaclass_branch.ctl" "baseml brown_f84_nhomo4.ctl" "baseml hiv2_tipdate.ctl" "baseml horai_mg4_malpha.ctl" "baseml brown_hky85_clock2.ctl" "codeml mtcdnapri_fromcodon.ctl" "codeml mtcdnapri_fromcodon0.ctl" "codeml mtcdnapri_aadist1.ctl"; do
  set -- $c
  t0=$(date +%s.%N)
  out=$(paml_amd/lib/pamlh_lnl $1 tests/golden/ctl/$2 --optimize 2>&1 | grep -E "lnL  =|ntime" | tr "\n" " ")
  t1=$(date +%s.%N)
  printf "%-36s %s  %s s\n" "$2" "$out" $(python3 -c "print(round($t1 - $t0, 2))")
done
