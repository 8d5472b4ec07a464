#!/bin/bash
# Where codeml_gpu (the reference's codeml with integration/*.patch) spends its time on the HIV NSsites = 0 2 example:
# wall time of three plain runs, then one run under rocprofv3 (HIP API + kernel statistics).  Output: gpurun_out/r04/binding_probe/
set -u
cd "$(dirname "$0")/.."
OUT=$PWD/gpurun_out/r04/binding_probe; mkdir -p $OUT
D=$(mktemp -d); cd $D
python - > codeml.ctl <<PY
import sys, os
sys.path[:0] = ["$OLDPWD/tests", "$OLDPWD"]
import test_reference_binding_gpu as t
print(t.HIV_CTL % {"data": t.DATA})
PY
BIN=$OLDPWD/oracle/_ref/codeml_gpu
TIMEFORMAT="%R s wall, %U user, %S sys"
for i in 1 2 3; do { time $BIN codeml.ctl < /dev/null > out$i.txt 2>&1; } 2>&1 | tail -1; done | tee $OUT/wall.txt
tail -3 out1.txt; grep "lnL(" mlc | tee -a $OUT/wall.txt
export TMPDIR=/tmp
rocprofv3 --hip-trace --kernel-trace --stats --output-format csv -d $OUT/prof -o hiv -- $BIN codeml.ctl < /dev/null > out_prof.txt 2>&1
ls $OUT/prof | head
find $OUT/prof -name "*stats*.csv" | while read f; do echo "== $f"; head -16 "$f" | cut -c1-180; done; find $OUT/prof -name "*trace.csv" -delete
