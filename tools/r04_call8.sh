#!/bin/bash
out=$PWD/gpurun_out/r04
mkdir -p $out
export TMPDIR=/tmp
timeout 900 python bench.py > $out/bench_1gpu.json 2>$out/bench_1gpu.err; echo "bench rc=$?"; tail -3 $out/bench_1gpu.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04/bench_1gpu.json"))
print({k: d[k] for k in ("value", "ms_per_step", "ms_per_step_readback")}, d["roofline"]["frac"], d["roofline"]["kernel_ms"])
print(json.dumps(d.get("fallbacks"), indent=0)[:3000])
b = d.get("branch"); print({k: b[k] for k in b if k != "roofline"} if isinstance(b, dict) else b); print(b.get("roofline") if isinstance(b, dict) else "")
print(d["c2"]["roofline"])
PY
timeout 1500 tools/collect_profiles_r04.sh > $out/collect_r04.log 2>&1; tail -60 $out/collect_r04.log
