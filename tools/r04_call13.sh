#!/bin/bash
out=$PWD/gpurun_out/r04
mkdir -p $out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu > $out/gpu_tests.txt 2>&1
grep "passed\|failed" $out/gpu_tests.txt | tail -2; grep -B25 "Error" $out/gpu_tests.txt | head -50
python __graft_entry__.py smoke 2>&1 | grep "^smoke"
timeout 900 python bench.py > $out/bench_1gpu.json 2>$out/bench_1gpu.err; echo "bench rc=$?"
timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $out/bench_driver_cmd.json 2>/dev/null
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04/bench_1gpu.json"))
print({k: d[k] for k in ("value", "ms_per_step", "ms_per_step_readback")}, d["roofline"]["frac"], d["roofline"]["kernel_ms"])
b = d.get("branch"); print({k: round(b[k], 4) for k in b if k.endswith("_ms")} if isinstance(b, dict) else b); print(b.get("roofline") if isinstance(b, dict) else "")
d2 = json.load(open("gpurun_out/r04/bench_driver_cmd.json")); print("driver cmd:", d2["value"], d2["ms_per_step"])
PY
