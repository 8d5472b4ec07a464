#!/bin/bash
# round 4: full GPU test tier, the bench line with the branch block, the exchange block against the device-mode stand-in, wgs = 2 / 4 emulation
out=$PWD/gpurun_out/r04
mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -x -q -m gpu > $out/gpu_tests.txt 2>&1
tail -4 $out/gpu_tests.txt
timeout 600 python bench.py > $out/bench_1gpu.json 2>$out/bench_1gpu.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04/bench_1gpu.json"))
print({k: d[k] for k in ("value", "ms_per_step", "ms_per_step_readback")}, d["roofline"]["frac"], d["roofline"]["kernel_ms"])
print(json.dumps(d.get("branch"), indent=0))
PY
PAML_AMD_RCCL_LIB=$PWD/tests/shim/librccl_shim.so PAML_AMD_SHIM_DEVICE_US=40 PAML_AMD_BENCH_FORCE_DIST=1 timeout 300 python bench.py --no-extras --no-cpu-baseline --patterns 124928 > $out/bench_exchange_shard8_T40.json 2>$out/bench_exchange.err
python -c "
import json; d=json.load(open('gpurun_out/r04/bench_exchange_shard8_T40.json')); print(d['ms_per_step'], d.get('exchange'))"
PAML_AMD_RCCL_LIB=$PWD/tests/shim/librccl_shim.so timeout 600 python tools/comm_emulated.py --steps 200 --ranks 8,1 --T 40,100 --wgs 2,4 --modes dual > $out/comm_emulated_wgs.txt 2>>$out/comm_emulated.err
grep -A12 "patterns  mode" $out/comm_emulated_wgs.txt
