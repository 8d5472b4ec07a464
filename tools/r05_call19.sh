#!/bin/bash
# round 5, GPU call 19: the whole `bench.py --gpus 8` line, self-launched, eight ranks on the one GPU through the stand-in for RCCL
O=gpurun_out/r05t; mkdir -p $O
export PAML_AMD_BENCH_ONE_GPU=1 PAML_AMD_RCCL_LIB=$PWD/tests/shim/librccl_shim.so PAML_AMD_BENCH_EXTRAS_S=200
s=$(date +%s); timeout 900 python bench.py --gpus 8 --steps 20 --warmup 5 > $O/bench8.json 2> $O/bench8.err; echo "rc=$? wall $(( $(date +%s) - s )) s"
tail -c 1500 $O/bench8.json; tail -5 $O/bench8.err
