#!/bin/bash
# where a step of the headline loop goes: kernel start / end times (rocprofv3 --kernel-trace) of a few consecutive evaluations
out=$PWD/gpurun_out/steptl; rm -rf $out; mkdir -p $out; export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $out -o t -- python $OLDPWD/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > /dev/null 2>&1)
python - "$out" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
ks = [(r["Kernel_Name"].split("(")[0][-24:], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Stream_Id", r.get("Queue_Id", ""))) for r in rows]
# the timed loop: the last 20 prune_jit before the profiled evaluations -> print evaluations 10..13
pj = [i for i, k in enumerate(ks) if k[0].endswith("prune_jit")]
i0 = pj[12]
t0 = ks[i0][1]
for k in ks[i0:i0 + 14]:
    print("%-26s start %9.1f us  end %9.1f us  dur %8.1f  stream/queue %s" % (k[0], (k[1] - t0) / 1e3, (k[2] - t0) / 1e3, (k[2] - k[1]) / 1e3, k[3]))
PY
rm -rf $out
