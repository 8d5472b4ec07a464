#!/bin/bash
# Round 6: the batched eigen-decomposition after its rewrite (2 x 2 blocks): its tests, the optimiser's view (tools/eigen_probe.py) and the
# kernel's own durations (rocprofv3 --kernel-trace --stats of the probe), for the default build and the variants given as arguments.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/eig
export TMPDIR=/tmp
REPO=$(pwd)
timeout 300 python -m pytest tests/test_eigen_gpu.py -q -x 2>&1 | tail -5
for lib in "" "$@"; do
  tag=$(basename "${lib:-default}" .so)
  echo "== $tag"
  PAML_AMD_LIB=$lib timeout 200 python tools/eigen_probe.py 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
for k, v in d.items(): print('%-40s %.3f ms  sweeps %.0f (max %d)' % (k, v['ms_median'], v['sweeps_median'], v['sweeps_max']))"
  [ -n "$lib" ] && lib=$REPO/$lib
  (cd /tmp && PAML_AMD_LIB=$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/eigprof_$tag -o eig -- python $REPO/tools/eigen_probe.py > /tmp/eigprof_$tag.log 2>&1) || tail -5 /tmp/eigprof_$tag.log
  f=$(find /tmp/eigprof_$tag -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp $f gpurun_out/eig/${tag}_kernel_stats.csv && grep -i "eigen" $f | head -3
  t=$(find /tmp/eigprof_$tag -name '*kernel_trace.csv' | head -1)
  [ -n "$t" ] && python - "$t" <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "eigen_qrev" in r["Kernel_Name"]]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
# the probe's order: cold b1 fd (40), cold b1 ls, cold b3 fd, cold b3 ls, warm ... (40 launches each)
names = ["cold b1 fd", "cold b1 ls", "cold b3 fd", "cold b3 ls", "warm b1 fd", "warm b1 ls", "warm b3 fd", "warm b3 ls"]
for i, nm in enumerate(names):
    seg = sorted(d[i * 40 + 5:(i + 1) * 40])
    if seg: print("kernel us  %-12s median %.1f  min %.1f  max %.1f" % (nm, seg[len(seg) // 2], seg[0], seg[-1]))
PY
done
