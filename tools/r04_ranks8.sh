#!/bin/bash
# the whole 8-rank path of bench.py on ONE GPU (every rank on GPU 0, gloo as the courier, the shared-memory stand-in for RCCL): not a
# performance figure — the lnL bits, the sweep, the weak block, the replicas, the exchange diagnostics, the guard
out=$PWD/gpurun_out/r04f; mkdir -p $out; export TMPDIR=/tmp
export PAML_AMD_BENCH_ONE_GPU=1 PAML_AMD_RCCL_LIB=$PWD/tests/shim/librccl_shim.so
timeout 420 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 8 --steps 5 --warmup 2 > $out/bench_8ranks_one_gpu.json 2> $out/bench_8ranks_one_gpu.err
echo rc=$?
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04f/bench_8ranks_one_gpu.json").read().strip().splitlines()[-1])
print("n_gpus", d["n_gpus"], "lnL_hex", d["lnL_hex"], "ms_per_step", d["ms_per_step"], "keys", sorted(k for k in d if k not in ("roofline", "config")))
print("exchange", {k: d["exchange"][k] for k in ("exchange_us", "lane_wait_us")} if "exchange" in d else None, "extras_error", d.get("extras_error"))
print("sweep lnL", [s["lnL"] for s in d.get("sweep", [])]); print("c5_replicas", d.get("c5_replicas", {}).get("seconds"))
PY
tail -3 $out/bench_8ranks_one_gpu.err
